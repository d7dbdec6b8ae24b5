// No-op drawing stand-ins (see core.hpp in this directory).
#pragma once
#include "core.hpp"

namespace cv {
template <typename P>
inline void circle(Mat &, P, int, const Scalar &, int = 1, int = 8, int = 0) {}
template <typename P>
inline void line(Mat &, P, P, const Scalar &, int = 1, int = 8, int = 0) {}
}  // namespace cv
