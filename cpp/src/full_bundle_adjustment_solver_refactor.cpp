// See the header.  Reference line numbers: core/full_bundle_adjustment_solver_refactor.cpp.
#include "core/full_bundle_adjustment_solver_refactor.h"

#include <stdexcept>

namespace visual_navigation {
namespace analytic_solver {

FullBundleAdjustmentSolverRefactor::FullBundleAdjustmentSolverRefactor() {}

void FullBundleAdjustmentSolverRefactor::Reset() { impl_.Reset(); }

void FullBundleAdjustmentSolverRefactor::RegisterCamera(const Index camera_id, const OptimizerCamera &camera) {  // :68-94
  _BA_Camera c;
  c.fx = camera.fx;
  c.fy = camera.fy;
  c.cx = camera.cx;
  c.cy = camera.cy;
  c.pose_this_to_cam0 = camera.camera_to_body_pose;  // same role: X_c = T * X_body (:758 vs full_...cpp:746)
  impl_.AddCamera(camera_id, c);
}

void FullBundleAdjustmentSolverRefactor::RegisterWorldToBodyPose(Pose *original_pose) {  // :96-111
  impl_.AddPose(original_pose);
}

void FullBundleAdjustmentSolverRefactor::RegisterWorldPoint(Point *original_point) {  // :113-126
  impl_.AddPoint(original_point);
}

void FullBundleAdjustmentSolverRefactor::MakePoseFixed(Pose *original_pose) { impl_.MakePoseFixed(original_pose); }
void FullBundleAdjustmentSolverRefactor::MakePointFixed(Point *original_point) { impl_.MakePointFixed(original_point); }

void FullBundleAdjustmentSolverRefactor::AddObservation(const Index camera_id, Pose *related_pose,
                                                        Point *related_point, const Pixel &pixel) {
  impl_.AddObservation(camera_id, related_pose, related_point, pixel);
}

bool FullBundleAdjustmentSolverRefactor::Solve(Options options, Summary *summary) {  // :640-1071
  if (options.solver_type == SolverType::LEVENBERG_MARQUARDT)
    impl_.SetGaussNewton(false);
  else if (options.solver_type == SolverType::GAUSS_NEWTON)
    impl_.SetGaussNewton(true);  // :976-982: every step accepted, lambda stays at initial_lambda
  else
    throw std::runtime_error(
        "FullBundleAdjustmentSolverRefactor::Solve: solver_type must be GAUSS_NEWTON or LEVENBERG_MARQUARDT");
  return impl_.Solve(options, summary);
}

bool FullBundleAdjustmentSolverRefactor::SolveByGradientDescent(Options, Summary *) {
  throw std::runtime_error("FullBundleAdjustmentSolverRefactor::SolveByGradientDescent is not provided by the MI355X path");
}

std::string FullBundleAdjustmentSolverRefactor::GetSolverStatistics() const { return impl_.GetSolverStatistics(); }

}  // namespace analytic_solver
}  // namespace visual_navigation
