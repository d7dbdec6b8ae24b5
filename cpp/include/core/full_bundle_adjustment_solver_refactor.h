// FullBundleAdjustmentSolverRefactor — host facade with the refactored names of
// the reference (core/full_bundle_adjustment_solver_refactor.h:36-136): the same
// device path as FullBundleAdjustmentSolver behind RegisterCamera /
// RegisterWorldToBodyPose / RegisterWorldPoint, plus the solver_type switch of
// its Solve (reference ..._refactor.cpp:944-982).  Reference test/
// test_ba_refactor.cpp compiles against this header unchanged.
#ifndef BA_FACADE_FULL_BUNDLE_ADJUSTMENT_SOLVER_REFACTOR_H_
#define BA_FACADE_FULL_BUNDLE_ADJUSTMENT_SOLVER_REFACTOR_H_

#include <string>
#include <vector>

#include "core/full_bundle_adjustment_solver.h"
#include "core/solver_option_and_summary.h"
#include "eigen3/Eigen/Dense"
#include "utility/timer.h"

namespace visual_navigation {
namespace analytic_solver {

using SolverNumeric = double;
using Index = int;
using Pixel = Eigen::Matrix<SolverNumeric, 2, 1>;
using Point = Eigen::Matrix<SolverNumeric, 3, 1>;
using Rotation3D = Eigen::Matrix<SolverNumeric, 3, 3>;
using Translation3D = Eigen::Matrix<SolverNumeric, 3, 1>;
using Pose = Eigen::Transform<SolverNumeric, 3, 1>;
using ErrorList = std::vector<SolverNumeric>;
using IndexList = std::vector<Index>;
using PixelList = std::vector<Pixel>;
using PointList = std::vector<Point>;

struct OptimizerCamera {
  OptimizerCamera() {}
  OptimizerCamera(const OptimizerCamera &camera)
      : fx(camera.fx), fy(camera.fy), cx(camera.cx), cy(camera.cy), camera_to_body_pose(camera.camera_to_body_pose) {}
  OptimizerCamera &operator=(const OptimizerCamera &camera) = default;
  SolverNumeric fx{0.0};
  SolverNumeric fy{0.0};
  SolverNumeric cx{0.0};
  SolverNumeric cy{0.0};
  Pose camera_to_body_pose;  // applied as X_camera = camera_to_body_pose * X_body (reference ..._refactor.cpp:758)
};

struct PointObservation {
  int related_camera_id{-1};
  Pose *related_pose{nullptr};
  Point *related_point{nullptr};
  Pixel pixel{-1.0, -1.0};
};

class FullBundleAdjustmentSolverRefactor {
 public:
  FullBundleAdjustmentSolverRefactor();

  void Reset();

  void RegisterCamera(const Index camera_id, const OptimizerCamera &camera);
  void RegisterWorldToBodyPose(Pose *original_pose);
  void RegisterWorldPoint(Point *original_point);

  void MakePoseFixed(Pose *original_pose);
  void MakePointFixed(Point *original_point);

  // solver_type LEVENBERG_MARQUARDT or GAUSS_NEWTON (the default of Options);
  // anything else throws std::runtime_error
  bool Solve(Options options, Summary *summary = nullptr);
  // reference ..._refactor.cpp:1073-1370: not on the MI355X path, throws
  bool SolveByGradientDescent(Options options, Summary *summary = nullptr);

  std::string GetSolverStatistics() const;

  void AddObservation(const Index camera_id, Pose *related_pose, Point *related_point, const Pixel &pixel);

  void SetDevice(int device_id) { impl_.SetDevice(device_id); }
  void SetVerbose(bool on) { impl_.SetVerbose(on); }

 private:
  FullBundleAdjustmentSolver impl_;
};

}  // namespace analytic_solver
}  // namespace visual_navigation
#endif
