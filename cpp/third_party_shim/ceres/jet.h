// Forward-mode dual numbers for the minimal Ceres-API stand-in (see ceres.h in
// this directory).  Written from scratch: value `a` plus N partial derivatives
// `v`; arithmetic and the elementary functions the cost functors of this
// repository use.
#ifndef BA_SHIM_CERES_JET_H_
#define BA_SHIM_CERES_JET_H_

#include <cmath>

namespace ceres {

template <typename T, int N>
struct Jet {
  T a;
  T v[N];
  Jet() : a(T(0)) {
    for (int i = 0; i < N; ++i) v[i] = T(0);
  }
  Jet(const T &value) : a(value) {  // NOLINT: implicit on purpose (T(1.0), mixed arithmetic)
    for (int i = 0; i < N; ++i) v[i] = T(0);
  }
  Jet(const T &value, int k) : a(value) {  // the k-th independent variable
    for (int i = 0; i < N; ++i) v[i] = T(i == k ? 1 : 0);
  }
  Jet &operator+=(const Jet &o) { return *this = *this + o; }
  Jet &operator-=(const Jet &o) { return *this = *this - o; }
  Jet &operator*=(const Jet &o) { return *this = *this * o; }
  Jet &operator/=(const Jet &o) { return *this = *this / o; }
};

template <typename T, int N>
Jet<T, N> operator-(const Jet<T, N> &f) {
  Jet<T, N> r;
  r.a = -f.a;
  for (int i = 0; i < N; ++i) r.v[i] = -f.v[i];
  return r;
}
template <typename T, int N>
Jet<T, N> operator+(const Jet<T, N> &f, const Jet<T, N> &g) {
  Jet<T, N> r;
  r.a = f.a + g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i];
  return r;
}
template <typename T, int N>
Jet<T, N> operator-(const Jet<T, N> &f, const Jet<T, N> &g) {
  Jet<T, N> r;
  r.a = f.a - g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i];
  return r;
}
template <typename T, int N>
Jet<T, N> operator*(const Jet<T, N> &f, const Jet<T, N> &g) {
  Jet<T, N> r;
  r.a = f.a * g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
template <typename T, int N>
Jet<T, N> operator/(const Jet<T, N> &f, const Jet<T, N> &g) {
  Jet<T, N> r;
  const T inv = T(1) / g.a;
  r.a = f.a * inv;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - r.a * g.v[i]) * inv;
  return r;
}
// mixed Jet / scalar forms
#define BA_SHIM_JET_MIXED(op)                                                        \
  template <typename T, int N>                                                       \
  Jet<T, N> operator op(const Jet<T, N> &f, const T &s) { return f op Jet<T, N>(s); } \
  template <typename T, int N>                                                       \
  Jet<T, N> operator op(const T &s, const Jet<T, N> &f) { return Jet<T, N>(s) op f; }
BA_SHIM_JET_MIXED(+)
BA_SHIM_JET_MIXED(-)
BA_SHIM_JET_MIXED(*)
BA_SHIM_JET_MIXED(/)
#undef BA_SHIM_JET_MIXED

#define BA_SHIM_JET_CMP(op)                                                        \
  template <typename T, int N>                                                     \
  bool operator op(const Jet<T, N> &f, const Jet<T, N> &g) { return f.a op g.a; }  \
  template <typename T, int N>                                                     \
  bool operator op(const Jet<T, N> &f, const T &s) { return f.a op s; }            \
  template <typename T, int N>                                                     \
  bool operator op(const T &s, const Jet<T, N> &f) { return s op f.a; }
BA_SHIM_JET_CMP(<)
BA_SHIM_JET_CMP(<=)
BA_SHIM_JET_CMP(>)
BA_SHIM_JET_CMP(>=)
BA_SHIM_JET_CMP(==)
BA_SHIM_JET_CMP(!=)
#undef BA_SHIM_JET_CMP

template <typename T, int N>
Jet<T, N> chain(const Jet<T, N> &f, const T &value, const T &derivative) {
  Jet<T, N> r;
  r.a = value;
  for (int i = 0; i < N; ++i) r.v[i] = derivative * f.v[i];
  return r;
}
template <typename T, int N>
Jet<T, N> sqrt(const Jet<T, N> &f) {
  const T s = std::sqrt(f.a);
  return chain(f, s, T(1) / (T(2) * s));
}
template <typename T, int N>
Jet<T, N> sin(const Jet<T, N> &f) { return chain(f, std::sin(f.a), std::cos(f.a)); }
template <typename T, int N>
Jet<T, N> cos(const Jet<T, N> &f) { return chain(f, std::cos(f.a), -std::sin(f.a)); }
template <typename T, int N>
Jet<T, N> exp(const Jet<T, N> &f) { const T e = std::exp(f.a); return chain(f, e, e); }
template <typename T, int N>
Jet<T, N> log(const Jet<T, N> &f) { return chain(f, std::log(f.a), T(1) / f.a); }
template <typename T, int N>
Jet<T, N> abs(const Jet<T, N> &f) { return f.a < T(0) ? -f : f; }
template <typename T, int N>
Jet<T, N> atan2(const Jet<T, N> &y, const Jet<T, N> &x) {
  const T d = x.a * x.a + y.a * y.a;
  Jet<T, N> r;
  r.a = std::atan2(y.a, x.a);
  for (int i = 0; i < N; ++i) r.v[i] = (x.a * y.v[i] - y.a * x.v[i]) / d;
  return r;
}
// scalar overloads so that templated functors can call ceres::sqrt etc. on double
using std::abs;
using std::atan2;
using std::cos;
using std::exp;
using std::log;
using std::sin;
using std::sqrt;

}  // namespace ceres
#endif
