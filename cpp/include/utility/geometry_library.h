// Lie-group helpers with the names and signatures callers of the reference use
// (reference utility/geometry_library.h:40-52): exponential and logarithm maps
// of SO(3) / SE(3), float ("_f") and double.  Twists are [v; w] (translation
// part first), as in reference utility/geometry_library.cpp:429-588.
// Re-derived here from the closed forms
//   exp(w^)  = I + A w^ + B w^2,           A = sin t / t, B = (1 - cos t) / t^2
//   V        = I + B w^ + C w^2,           C = (t - sin t) / t^3,  t_SE3 = V v
//   V^-1     = I - w^/2 + (1 - A / (2B)) / t^2 * w^2
// with series limits for t -> 0.  Header-only.
#ifndef BA_FACADE_GEOMETRY_LIBRARY_H_
#define BA_FACADE_GEOMETRY_LIBRARY_H_

#include <cmath>
#include <iostream>

#include "eigen3/Eigen/Dense"

using namespace Eigen;  // the reference header does the same at global scope (:8)

namespace geometry {

namespace detail {
template <typename T>
Eigen::Matrix<T, 3, 3> Hat(const Eigen::Matrix<T, 3, 1> &w) {
  Eigen::Matrix<T, 3, 3> m;
  m(0, 1) = -w(2); m(0, 2) = w(1);
  m(1, 0) = w(2);  m(1, 2) = -w(0);
  m(2, 0) = -w(1); m(2, 1) = w(0);
  return m;
}
template <typename T>
void So3Exp(const Eigen::Matrix<T, 3, 1> &w, Eigen::Matrix<T, 3, 3> &R) {
  const T t = std::sqrt(w.dot(w));
  const Eigen::Matrix<T, 3, 3> K = Hat(w), K2 = K * K;
  const T A = t < T(1e-9) ? T(1) : std::sin(t) / t;
  const T B = t < T(1e-9) ? T(0.5) : (T(1) - std::cos(t)) / (t * t);
  R = Eigen::Matrix<T, 3, 3>::Identity() + K * A + K2 * B;
}
template <typename T>
void So3Log(const Eigen::Matrix<T, 3, 3> &R, Eigen::Matrix<T, 3, 1> &w) {
  T c = (R(0, 0) + R(1, 1) + R(2, 2) - T(1)) * T(0.5);
  c = c > T(1) ? T(1) : (c < T(-1) ? T(-1) : c);
  const T t = std::acos(c);
  const T k = t < T(1e-9) ? T(0.5) : t / (T(2) * std::sin(t));
  w(0) = k * (R(2, 1) - R(1, 2));
  w(1) = k * (R(0, 2) - R(2, 0));
  w(2) = k * (R(1, 0) - R(0, 1));
}
template <typename T>
void Se3Exp(const Eigen::Matrix<T, 6, 1> &xi, Eigen::Matrix<T, 4, 4> &Tm) {
  const Eigen::Matrix<T, 3, 1> v(xi(0), xi(1), xi(2)), w(xi(3), xi(4), xi(5));
  const T t = std::sqrt(w.dot(w));
  const Eigen::Matrix<T, 3, 3> K = Hat(w), K2 = K * K;
  Eigen::Matrix<T, 3, 3> R;
  So3Exp(w, R);
  const T B = t < T(1e-9) ? T(0.5) : (T(1) - std::cos(t)) / (t * t);
  const T C = t < T(1e-9) ? T(1) / T(6) : (t - std::sin(t)) / (t * t * t);
  const Eigen::Matrix<T, 3, 1> tr = (Eigen::Matrix<T, 3, 3>::Identity() + K * B + K2 * C) * v;
  Tm = Eigen::Matrix<T, 4, 4>::Identity();
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Tm(r, c) = R(r, c);
    Tm(r, 3) = tr(r);
  }
}
template <typename T>
void Se3Log(const Eigen::Matrix<T, 4, 4> &Tm, Eigen::Matrix<T, 6, 1> &xi) {
  Eigen::Matrix<T, 3, 3> R;
  Eigen::Matrix<T, 3, 1> tr, w;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) R(r, c) = Tm(r, c);
    tr(r) = Tm(r, 3);
  }
  So3Log(R, w);
  const T t = std::sqrt(w.dot(w));
  Eigen::Matrix<T, 3, 3> Vinv = Eigen::Matrix<T, 3, 3>::Identity();
  if (t >= T(1e-9)) {
    const Eigen::Matrix<T, 3, 3> K = Hat(w);
    const T A = std::sin(t) / t, B = (T(1) - std::cos(t)) / (t * t);
    Vinv = Vinv - K * T(0.5) + (K * K) * ((T(1) - A / (T(2) * B)) / (t * t));
  }
  const Eigen::Matrix<T, 3, 1> v = Vinv * tr;
  for (int k = 0; k < 3; ++k) {
    xi(k) = v(k);
    xi(3 + k) = w(k);
  }
}
}  // namespace detail

inline Matrix3d skewMat(const Vector3d &v) { return detail::Hat(v); }
inline Matrix3f skewMat_f(const Vector3f &v) { return detail::Hat(v); }

inline void se3Exp(const Eigen::Matrix<double, 6, 1> &xi, Eigen::Matrix<double, 4, 4> &T) { detail::Se3Exp(xi, T); }
inline void se3Exp_f(const Eigen::Matrix<float, 6, 1> &xi, Eigen::Matrix4f &T) { detail::Se3Exp(xi, T); }
inline void SE3Log(const Eigen::Matrix<double, 4, 4> &T, Eigen::Matrix<double, 6, 1> &xi) { detail::Se3Log(T, xi); }
inline void SE3Log_f(const Eigen::Matrix<float, 4, 4> &T, Eigen::Matrix<float, 6, 1> &xi) { detail::Se3Log(T, xi); }

inline void so3Exp(const Eigen::Matrix<double, 3, 1> &w, Eigen::Matrix<double, 3, 3> &R) { detail::So3Exp(w, R); }
inline void so3Exp(const double w1, const double w2, const double w3, Eigen::Matrix<double, 3, 3> &R) {
  detail::So3Exp(Eigen::Matrix<double, 3, 1>(w1, w2, w3), R);
}
inline void so3Exp_f(const Eigen::Matrix<float, 3, 1> &w, Eigen::Matrix3f &R) { detail::So3Exp(w, R); }
inline void SO3Log(const Eigen::Matrix<double, 3, 3> &R, Eigen::Matrix<double, 3, 1> &w) { detail::So3Log(R, w); }
inline void SO3Log_f(const Eigen::Matrix<float, 3, 3> &R, Eigen::Matrix<float, 3, 1> &w) { detail::So3Log(R, w); }

inline Eigen::Matrix<double, 4, 4> inverseSE3(const Eigen::Matrix<double, 4, 4> &T) {
  Eigen::Matrix<double, 4, 4> o = Eigen::Matrix<double, 4, 4>::Identity();
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o(r, c) = T(c, r);
    o(r, 3) = -(T(0, r) * T(0, 3) + T(1, r) * T(1, 3) + T(2, r) * T(2, 3));
  }
  return o;
}
inline Eigen::Matrix<float, 4, 4> inverseSE3_f(const Eigen::Matrix<float, 4, 4> &T) {
  Eigen::Matrix<float, 4, 4> o = Eigen::Matrix<float, 4, 4>::Identity();
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o(r, c) = T(c, r);
    o(r, 3) = -(T(0, r) * T(0, 3) + T(1, r) * T(1, 3) + T(2, r) * T(2, 3));
  }
  return o;
}

// xi <- log(exp(dxi) exp(xi)): left-multiplicative update of a twist
inline void addFrontse3(Eigen::Matrix<double, 6, 1> &xi, const Eigen::Matrix<double, 6, 1> &dxi) {
  Eigen::Matrix<double, 4, 4> T, dT;
  se3Exp(xi, T);
  se3Exp(dxi, dT);
  SE3Log(dT * T, xi);
}
inline void addFrontse3_f(Eigen::Matrix<float, 6, 1> &xi, const Eigen::Matrix<float, 6, 1> &dxi) {
  Eigen::Matrix<float, 4, 4> T, dT;
  se3Exp_f(xi, T);
  se3Exp_f(dxi, dT);
  SE3Log_f(dT * T, xi);
}

}  // namespace geometry
#endif
