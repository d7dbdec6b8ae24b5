// Reference: core/pose_only_bundle_adjustment_solver.cpp:8-170 (monocular), :172-399 (stereo).
#include "core/pose_only_bundle_adjustment_solver.h"

#include <stdexcept>

#include "ba_hip.h"
#include "utility/timer.h"

namespace visual_navigation {
namespace analytic_solver {

PoseOnlyBundleAdjustmentSolver::PoseOnlyBundleAdjustmentSolver() {}
PoseOnlyBundleAdjustmentSolver::~PoseOnlyBundleAdjustmentSolver() {
  if (handle_) ba_destroy(handle_);
}

const std::vector<Eigen::Isometry3f> &PoseOnlyBundleAdjustmentSolver::GetDebugPoses() const { return debug_poses_; }

bool PoseOnlyBundleAdjustmentSolver::Solve_Monocular_6Dof(
    const std::vector<Eigen::Vector3f> &reference_position_list,
    const std::vector<Eigen::Vector2f> &matched_pixel_list, const float fx, const float fy, const float cx,
    const float cy, Eigen::Isometry3f &reference_to_current_pose, std::vector<bool> &mask_inlier, Options options,
    Summary *summary) {
  timer::StopWatch stopwatch("SolveMonocularPoseOnlyBundleAdjustment6Dof");
  stopwatch.Start();
  if (summary != nullptr) {
    summary->max_iteration_ = options.iteration_handle.max_num_iterations;
    summary->threshold_cost_change_ = options.convergence_handle.threshold_cost_change;
    summary->threshold_step_size_ = options.convergence_handle.threshold_step_size;
    summary->convergence_status_ = true;
  }
  debug_poses_.resize(0);
  if (reference_position_list.size() != matched_pixel_list.size())
    throw std::runtime_error(
        "In PoseOnlyBundleAdjustmentSolver::SolveMonocularPoseOnlyBundleAdjustment6Dof(), "
        "world_position_list.size() != current_pixel_list.size()");
  const int n = static_cast<int>(reference_position_list.size());
  mask_inlier.resize(n, true);
  if (n == 0) return true;
  if (!handle_ && ba_create(&handle_, 0) < 0) throw std::runtime_error(ba_last_error());

  std::vector<float> X(3 * n), uv(2 * n);
  std::vector<uint8_t> mask(n);
  for (int k = 0; k < n; ++k) {
    for (int r = 0; r < 3; ++r) X[3 * k + r] = reference_position_list[k](r);
    uv[2 * k] = matched_pixel_list[k](0);
    uv[2 * k + 1] = matched_pixel_list[k](1);
    mask[k] = mask_inlier[k] ? 1 : 0;
  }
  float T12[12];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T12[3 * r + c] = reference_to_current_pose.linear()(r, c);
    T12[9 + r] = reference_to_current_pose.translation()(r);
  }
  ba_options o;
  o.threshold_step_size = options.convergence_handle.threshold_step_size;
  o.threshold_cost_change = options.convergence_handle.threshold_cost_change;
  o.threshold_huber_loss = options.outlier_handle.threshold_huber_loss;
  o.threshold_outlier_rejection = options.outlier_handle.threshold_outlier_rejection;
  o.max_num_iterations = options.iteration_handle.max_num_iterations;
  o.initial_lambda = options.trust_region_handle.initial_lambda;
  o.decrease_ratio_lambda = options.trust_region_handle.decrease_ratio_lambda;
  o.increase_ratio_lambda = options.trust_region_handle.increase_ratio_lambda;
  o.gauss_newton = 0;
  const int cap = o.max_num_iterations > 0 ? o.max_num_iterations : 1;
  std::vector<ba_po_iter> rows(cap);
  std::vector<float> dbg(static_cast<size_t>(cap) * 12);
  int n_iter = 0, converged = 0;
  const int rc = ba_pose_only_mono6(handle_, X.data(), uv.data(), n, fx, fy, cx, cy, T12, mask.data(), &o,
                                    rows.data(), cap, &n_iter, &converged, dbg.data());
  if (rc < 0) throw std::runtime_error(ba_last_error());
  for (int k = 0; k < n; ++k) mask_inlier[k] = mask[k] != 0;
  for (int it = 0; it < n_iter && it < cap; ++it) {
    Eigen::Isometry3f D;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) D.linear()(r, c) = dbg[12 * it + 3 * r + c];
      D.translation()(r) = dbg[12 * it + 9 + r];
    }
    debug_poses_.push_back(D);
  }
  const bool is_success = (rc == 0);
  if (is_success) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) reference_to_current_pose.linear()(r, c) = T12[3 * r + c];
      reference_to_current_pose.translation()(r) = T12[9 + r];
    }
  } else {
    std::cout << "!! WARNING !! poseonly BA yields NAN value!!\n";
  }
  if (summary != nullptr) {
    const int n_rows = converged ? n_iter - 1 : n_iter;  // no row on the converging iteration (:116-121)
    for (int k = 0; k < n_rows && k < cap; ++k) {
      OptimizationInfo info;
      info.cost = rows[k].cost;
      info.cost_change = rows[k].cost_change;
      info.average_reprojection_error = rows[k].cost;
      info.abs_step = rows[k].abs_step;
      info.abs_gradient = 0;
      info.damping_term = -1;
      info.iter_time = 0.0;
      info.iteration_status = IterationStatus::UPDATE;
      summary->optimization_info_list_.push_back(info);
    }
    summary->convergence_status_ = converged != 0;
    summary->total_time_in_millisecond_ = stopwatch.GetLapTimeFromStart();
  }
  return is_success;
}

bool PoseOnlyBundleAdjustmentSolver::Solve_Stereo_6Dof(
    const std::vector<Eigen::Vector3f> &reference_position_list,
    const std::vector<Eigen::Vector2f> &matched_left_pixel_list,
    const std::vector<Eigen::Vector2f> &matched_right_pixel_list, const float fx_left, const float fy_left,
    const float cx_left, const float cy_left, const float fx_right, const float fy_right, const float cx_right,
    const float cy_right, const Eigen::Isometry3f &left_to_right_pose,
    Eigen::Isometry3f &reference_to_current_left_pose, std::vector<bool> &mask_inlier_left,
    std::vector<bool> &mask_inlier_right, Options options, Summary *summary) {
  timer::StopWatch stopwatch("SolveStereoPoseOnlyBundleAdjustment6Dof");
  stopwatch.Start();
  if (summary != nullptr) {
    summary->max_iteration_ = options.iteration_handle.max_num_iterations;
    summary->threshold_cost_change_ = options.convergence_handle.threshold_cost_change;
    summary->threshold_step_size_ = options.convergence_handle.threshold_step_size;
    summary->convergence_status_ = true;
  }
  debug_poses_.resize(0);
  if (reference_position_list.size() != matched_left_pixel_list.size())  // :203-208
    throw std::runtime_error(
        "In PoseOnlyBundleAdjustmentSolver::SolveStereoPoseOnlyBundleAdjustment6Dof(), "
        "world_position_list.size() != left_current_pixel_list.size()");
  if (reference_position_list.size() != matched_right_pixel_list.size())  // :209-214
    throw std::runtime_error(
        "In PoseOnlyBundleAdjustmentSolver::SolveStereoPoseOnlyBundleAdjustment6Dof(), "
        "world_position_list.size() != right_current_pixel_list.size()");
  const int n = static_cast<int>(reference_position_list.size());
  mask_inlier_left.resize(n, true);
  mask_inlier_right.resize(n, true);
  if (n == 0) return true;
  if (!handle_ && ba_create(&handle_, 0) < 0) throw std::runtime_error(ba_last_error());

  std::vector<float> X(3 * n), uvl(2 * n), uvr(2 * n);
  std::vector<uint8_t> ml(n), mr(n);
  for (int k = 0; k < n; ++k) {
    for (int r = 0; r < 3; ++r) X[3 * k + r] = reference_position_list[k](r);
    uvl[2 * k] = matched_left_pixel_list[k](0);
    uvl[2 * k + 1] = matched_left_pixel_list[k](1);
    uvr[2 * k] = matched_right_pixel_list[k](0);
    uvr[2 * k + 1] = matched_right_pixel_list[k](1);
    ml[k] = mask_inlier_left[k] ? 1 : 0;
    mr[k] = mask_inlier_right[k] ? 1 : 0;
  }
  float T12[12], Tlr[12];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      T12[3 * r + c] = reference_to_current_left_pose.linear()(r, c);
      Tlr[3 * r + c] = left_to_right_pose.linear()(r, c);
    }
    T12[9 + r] = reference_to_current_left_pose.translation()(r);
    Tlr[9 + r] = left_to_right_pose.translation()(r);
  }
  const float il[4] = {fx_left, fy_left, cx_left, cy_left};
  const float ir[4] = {fx_right, fy_right, cx_right, cy_right};
  ba_options o;
  o.threshold_step_size = options.convergence_handle.threshold_step_size;
  o.threshold_cost_change = options.convergence_handle.threshold_cost_change;
  o.threshold_huber_loss = options.outlier_handle.threshold_huber_loss;
  o.threshold_outlier_rejection = options.outlier_handle.threshold_outlier_rejection;
  o.max_num_iterations = options.iteration_handle.max_num_iterations;
  o.initial_lambda = options.trust_region_handle.initial_lambda;
  o.decrease_ratio_lambda = options.trust_region_handle.decrease_ratio_lambda;
  o.increase_ratio_lambda = options.trust_region_handle.increase_ratio_lambda;
  o.gauss_newton = 0;
  const int cap = o.max_num_iterations > 0 ? o.max_num_iterations : 1;
  std::vector<ba_po_iter> rows(cap);
  std::vector<float> dbg(static_cast<size_t>(cap) * 12);
  int n_iter = 0, converged = 0;
  const int rc = ba_pose_only_stereo6(handle_, X.data(), uvl.data(), uvr.data(), n, il, ir, Tlr, T12, ml.data(),
                                      mr.data(), &o, rows.data(), cap, &n_iter, &converged, dbg.data());
  if (rc < 0) throw std::runtime_error(ba_last_error());
  for (int k = 0; k < n; ++k) {
    mask_inlier_left[k] = ml[k] != 0;
    mask_inlier_right[k] = mr[k] != 0;
  }
  for (int it = 0; it < n_iter && it < cap; ++it) {
    Eigen::Isometry3f D;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) D.linear()(r, c) = dbg[12 * it + 3 * r + c];
      D.translation()(r) = dbg[12 * it + 9 + r];
    }
    debug_poses_.push_back(D);
  }
  const bool is_success = (rc == 0);
  if (is_success) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) reference_to_current_left_pose.linear()(r, c) = T12[3 * r + c];
      reference_to_current_left_pose.translation()(r) = T12[9 + r];
    }
  } else {
    std::cout << "!! WARNING !! poseonly BA yields NAN value!!\n";
  }
  if (summary != nullptr) {
    const int n_rows = converged ? n_iter - 1 : n_iter;  // no row on the converging iteration
    for (int k = 0; k < n_rows && k < cap; ++k) {
      OptimizationInfo info;
      info.cost = rows[k].cost;
      info.cost_change = rows[k].cost_change;
      info.average_reprojection_error = rows[k].cost;
      info.abs_step = rows[k].abs_step;
      info.abs_gradient = 0;
      info.damping_term = -1;
      info.iter_time = 0.0;
      info.iteration_status = IterationStatus::UPDATE;
      summary->optimization_info_list_.push_back(info);
    }
    summary->convergence_status_ = converged != 0;
    summary->total_time_in_millisecond_ = stopwatch.GetLapTimeFromStart();
  }
  return is_success;
}

}  // namespace analytic_solver
}  // namespace visual_navigation
