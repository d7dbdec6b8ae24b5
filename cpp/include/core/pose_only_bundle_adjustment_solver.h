// PoseOnlyBundleAdjustmentSolver — C++ facade of the pose-only path over the
// HIP C ABI.  Mirrors the monocular 6-DoF entry point of the reference
// (core/pose_only_bundle_adjustment_solver.h:25-67); the planar 3-DoF and
// stereo variants are out of scope this round.
#ifndef BA_FACADE_POSE_ONLY_BUNDLE_ADJUSTMENT_SOLVER_H_
#define BA_FACADE_POSE_ONLY_BUNDLE_ADJUSTMENT_SOLVER_H_

#include <vector>

#include "core/solver_option_and_summary.h"
#include "eigen3/Eigen/Dense"
#include "eigen3/Eigen/Geometry"

struct ba_handle;

namespace visual_navigation {
namespace analytic_solver {

class PoseOnlyBundleAdjustmentSolver {
 public:
  PoseOnlyBundleAdjustmentSolver();
  ~PoseOnlyBundleAdjustmentSolver();
  PoseOnlyBundleAdjustmentSolver(const PoseOnlyBundleAdjustmentSolver &) = delete;

  bool Solve_Monocular_6Dof(const std::vector<Eigen::Vector3f> &reference_position_list,
                            const std::vector<Eigen::Vector2f> &matched_pixel_list, const float fx, const float fy,
                            const float cx, const float cy, Eigen::Isometry3f &reference_to_current_pose,
                            std::vector<bool> &mask_inlier, Options options, Summary *summary = nullptr);

  // reference core/pose_only_bundle_adjustment_solver.h (Solve_Stereo_6Dof), .cpp:172-399
  bool Solve_Stereo_6Dof(const std::vector<Eigen::Vector3f> &reference_position_list,
                         const std::vector<Eigen::Vector2f> &matched_left_pixel_list,
                         const std::vector<Eigen::Vector2f> &matched_right_pixel_list, const float fx_left,
                         const float fy_left, const float cx_left, const float cy_left, const float fx_right,
                         const float fy_right, const float cx_right, const float cy_right,
                         const Eigen::Isometry3f &left_to_right_pose,
                         Eigen::Isometry3f &reference_to_current_left_pose, std::vector<bool> &mask_inlier_left,
                         std::vector<bool> &mask_inlier_right, Options options, Summary *summary = nullptr);

  const std::vector<Eigen::Isometry3f> &GetDebugPoses() const;

 private:
  ba_handle *handle_{nullptr};
  std::vector<Eigen::Isometry3f> debug_poses_;
};

}  // namespace analytic_solver
}  // namespace visual_navigation
#endif
