// Reprojection cost functors for Ceres-style automatic differentiation — the
// interface of reference core/pose_only_bundle_adjustment_solver_ceres.h:84-128
// (class names, constructor, static intrinsics setters, templated operator()),
// so that reference test/test_compare_ceres_vs_native.cpp compiles against this
// header unchanged.  Header-only (C++17 inline statics).  They are the
// independent fp64 check of the HIP pose-only solver: a world point is rotated
// by an ANGLE-AXIS vector, translated, projected; derivatives come from dual
// numbers, not from the analytic Jacobians of the solver under test.
//   6-DoF parameters: [wx wy wz tx ty tz]  (world -> camera)
//   3-DoF parameters: [tx ty yaw]          (planar base motion, camera mounted
//                                           on the base by a fixed extrinsic)
#ifndef BA_FACADE_POSE_ONLY_BUNDLE_ADJUSTMENT_SOLVER_CERES_H_
#define BA_FACADE_POSE_ONLY_BUNDLE_ADJUSTMENT_SOLVER_CERES_H_

#include <iostream>
#include <vector>

#include "ceres/ceres.h"
#include "ceres/rotation.h"
#include "eigen3/Eigen/Dense"
#include "eigen3/Eigen/Geometry"

class ReprojectionCostFunctor_6dof_numerical {
 public:
  ReprojectionCostFunctor_6dof_numerical() = delete;
  ReprojectionCostFunctor_6dof_numerical(const Eigen::Vector3d &world_position, const Eigen::Vector2d &pixel_matched)
      : world_position_(world_position), pixel_matched_(pixel_matched) {}

  template <typename T>
  bool operator()(const T *const params, T *residuals) const {
    const T X[3] = {T(world_position_(0)), T(world_position_(1)), T(world_position_(2))};
    T Xc[3];
    ceres::AngleAxisRotatePoint(params, X, Xc);
    for (int k = 0; k < 3; ++k) Xc[k] = Xc[k] + params[3 + k];
    const T iz = T(1.0) / Xc[2];
    residuals[0] = fx_ * Xc[0] * iz + cx_ - pixel_matched_(0);
    residuals[1] = fy_ * Xc[1] * iz + cy_ - pixel_matched_(1);
    return true;
  }

  // shared by every functor instance: call before building the problem
  static void SetCameraIntrinsicParameters(const double fx, const double fy, const double cx, const double cy) {
    fx_ = fx;
    fy_ = fy;
    cx_ = cx;
    cy_ = cy;
  }
  inline static double fx_ = 0.0;
  inline static double fy_ = 0.0;
  inline static double cx_ = 0.0;
  inline static double cy_ = 0.0;

 private:
  Eigen::Vector3d world_position_;
  Eigen::Vector2d pixel_matched_;
};

class ReprojectionCostFunctor_3dof_numerical {
 public:
  ReprojectionCostFunctor_3dof_numerical() = delete;
  ReprojectionCostFunctor_3dof_numerical(const Eigen::Vector3d &world_position, const Eigen::Vector2d &pixel_matched)
      : world_position_(world_position), pixel_matched_(pixel_matched) {}

  template <typename T>
  bool operator()(const T *const relative_baselink_parameter, T *residuals) const {
    const T c = cos(relative_baselink_parameter[2]), s = sin(relative_baselink_parameter[2]);
    // planar motion of the base, then the fixed base -> camera mounting
    const T Xb[3] = {c * world_position_(0) - s * world_position_(1) + relative_baselink_parameter[0],
                     s * world_position_(0) + c * world_position_(1) + relative_baselink_parameter[1],
                     T(world_position_(2))};
    const Eigen::Matrix3d &R = pose_camera_to_base_.linear();
    const Eigen::Vector3d &t = pose_camera_to_base_.translation();
    T Xc[3];
    for (int r = 0; r < 3; ++r) Xc[r] = R(r, 0) * Xb[0] + R(r, 1) * Xb[1] + R(r, 2) * Xb[2] + t(r);
    const T iz = T(1.0) / Xc[2];
    residuals[0] = fx_ * Xc[0] * iz + cx_ - pixel_matched_(0);
    residuals[1] = fy_ * Xc[1] * iz + cy_ - pixel_matched_(1);
    return true;
  }

  static void SetPoseBaseToCamera(const Eigen::Isometry3d &pose_base_to_camera) {
    pose_base_to_camera_ = pose_base_to_camera;
    pose_camera_to_base_ = pose_base_to_camera.inverse();
  }
  static void SetCameraIntrinsicParameters(const double fx, const double fy, const double cx, const double cy) {
    fx_ = fx;
    fy_ = fy;
    cx_ = cx;
    cy_ = cy;
  }
  inline static Eigen::Isometry3d pose_base_to_camera_ = Eigen::Isometry3d::Identity();
  inline static Eigen::Isometry3d pose_camera_to_base_ = Eigen::Isometry3d::Identity();
  inline static double fx_ = 0.0;
  inline static double fy_ = 0.0;
  inline static double cx_ = 0.0;
  inline static double cy_ = 0.0;

 private:
  Eigen::Vector3d world_position_;
  Eigen::Vector2d pixel_matched_;
};

#endif
