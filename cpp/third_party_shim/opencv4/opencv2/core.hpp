// No-op stand-in for the handful of OpenCV types the reference's demo programs
// touch (reference test/test_compare_ceres_vs_native.cpp:277-307 draws the
// optimisation progress; test/test_ba.cpp:9-11 only includes the headers).
// Used only where OpenCV is not installed (this image, the GPU box): images are
// size-only objects, drawing and display do nothing.
#pragma once

#define CV_8UC1 0
#define CV_8UC3 16

namespace cv {
struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
};
template <typename T>
struct Point_ {
  T x = T(0), y = T(0);
  Point_() {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
struct Scalar {
  double val[4];
  Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {}
};
class Mat {
 public:
  int rows = 0, cols = 0, type_ = 0;
  Mat() {}
  Mat(int r, int c, int t) : rows(r), cols(c), type_(t) {}
  static Mat zeros(Size s, int type) { return Mat(s.height, s.width, type); }
  static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
  Size size() const { return Size(cols, rows); }
  bool empty() const { return rows == 0 || cols == 0; }
};
}  // namespace cv
