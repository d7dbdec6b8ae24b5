// No-op display stand-ins (see core.hpp in this directory): nothing is shown
// and waitKey returns at once, so a demo that pauses on cv::waitKey(0)
// (reference test/test_compare_ceres_vs_native.cpp:306) runs through.
#pragma once
#include <string>

#include "core.hpp"

namespace cv {
inline void imshow(const std::string &, const Mat &) {}
inline int waitKey(int = 0) { return -1; }
inline void destroyAllWindows() {}
}  // namespace cv
