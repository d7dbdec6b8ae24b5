// ceres/rotation.h of the minimal Ceres-API stand-in (see ceres.h): the one
// function the pose-only cost functor needs.
#ifndef BA_SHIM_CERES_ROTATION_H_
#define BA_SHIM_CERES_ROTATION_H_

#include <limits>

#include "jet.h"

namespace ceres {

// result = R(angle_axis) * pt by Rodrigues' formula
//   p cos(t) + (w x p) sin(t) + w (w . p) (1 - cos(t)),  w = angle_axis / t,
// with the first-order form p + angle_axis x p near t = 0 so that the
// derivatives stay finite there.  T is double or a Jet.
template <typename T>
inline void AngleAxisRotatePoint(const T angle_axis[3], const T pt[3], T result[3]) {
  const T theta2 = angle_axis[0] * angle_axis[0] + angle_axis[1] * angle_axis[1] +
                   angle_axis[2] * angle_axis[2];
  if (theta2 > T(std::numeric_limits<double>::epsilon())) {
    const T theta = sqrt(theta2);
    const T c = cos(theta), s = sin(theta);
    const T inv = T(1.0) / theta;
    const T w[3] = {angle_axis[0] * inv, angle_axis[1] * inv, angle_axis[2] * inv};
    const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                      w[0] * pt[1] - w[1] * pt[0]};
    const T k = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - c);
    T out[3];
    for (int i = 0; i < 3; ++i) out[i] = pt[i] * c + wxp[i] * s + w[i] * k;
    for (int i = 0; i < 3; ++i) result[i] = out[i];
  } else {
    const T wxp[3] = {angle_axis[1] * pt[2] - angle_axis[2] * pt[1],
                      angle_axis[2] * pt[0] - angle_axis[0] * pt[2],
                      angle_axis[0] * pt[1] - angle_axis[1] * pt[0]};
    T out[3];
    for (int i = 0; i < 3; ++i) out[i] = pt[i] + wxp[i];
    for (int i = 0; i < 3; ++i) result[i] = out[i];
  }
}

}  // namespace ceres
#endif
