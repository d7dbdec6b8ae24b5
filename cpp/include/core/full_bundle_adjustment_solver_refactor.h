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

// scalar / vector aliases of the refactored API (same names, reference :36-51)
typedef double SolverNumeric;
typedef int Index;
typedef Eigen::Matrix<SolverNumeric, 2, 1> Pixel;          // image point [pixel]
typedef Eigen::Matrix<SolverNumeric, 3, 1> Point;          // world point [m]
typedef Eigen::Matrix<SolverNumeric, 3, 3> Rotation3D;
typedef Eigen::Matrix<SolverNumeric, 3, 1> Translation3D;
typedef Eigen::Transform<SolverNumeric, 3, 1> Pose;        // rigid (Isometry) transform
typedef std::vector<SolverNumeric> ErrorList;
typedef std::vector<Index> IndexList;
typedef std::vector<Pixel> PixelList;
typedef std::vector<Point> PointList;

// Pinhole camera of the rig.  camera_to_body_pose is applied as
// X_camera = camera_to_body_pose * X_body (reference ..._refactor.cpp:758).
struct OptimizerCamera {
  OptimizerCamera() = default;
  OptimizerCamera(const OptimizerCamera &other) = default;
  OptimizerCamera &operator=(const OptimizerCamera &other) = default;
  SolverNumeric fx = 0.0, fy = 0.0;  // focal lengths [pixel]
  SolverNumeric cx = 0.0, cy = 0.0;  // principal point [pixel]
  Pose camera_to_body_pose;
};

// One pixel measurement of a point from a pose through a camera.
struct PointObservation {
  int related_camera_id = -1;
  Pose *related_pose = nullptr;
  Point *related_point = nullptr;
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
