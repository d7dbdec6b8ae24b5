// FullBundleAdjustmentSolver — C++ host facade over the HIP C ABI
// (include/ba_hip.h).  Same namespace, class name, typedef names and method
// signatures as the reference (core/full_bundle_adjustment_solver.h:34-146),
// so a caller such as reference test/test_ba.cpp compiles against this header
// unchanged.  What the facade does on the host is exactly what the
// reference's Add* methods do (pointer -> index maps, 0.01 scaling, pose
// inversion, write-back); the whole LM loop runs on the GPU.
#ifndef BA_FACADE_FULL_BUNDLE_ADJUSTMENT_SOLVER_H_
#define BA_FACADE_FULL_BUNDLE_ADJUSTMENT_SOLVER_H_

#include <cstdint>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "core/solver_option_and_summary.h"
#include "eigen3/Eigen/Dense"
#include "utility/timer.h"

struct ba_handle;  // include/ba_hip.h

namespace visual_navigation {
namespace analytic_solver {

using _BA_Numeric = double;
using _BA_Index = int;
using _BA_Size_t = int;
using _BA_Mat33 = Eigen::Matrix<_BA_Numeric, 3, 3>;
using _BA_Mat66 = Eigen::Matrix<_BA_Numeric, 6, 6>;
using _BA_Vec2 = Eigen::Matrix<_BA_Numeric, 2, 1>;
using _BA_Vec3 = Eigen::Matrix<_BA_Numeric, 3, 1>;
using _BA_Vec6 = Eigen::Matrix<_BA_Numeric, 6, 1>;
using _BA_Pixel = Eigen::Matrix<_BA_Numeric, 2, 1>;
using _BA_Point = Eigen::Matrix<_BA_Numeric, 3, 1>;
using _BA_Rotation3 = Eigen::Matrix<_BA_Numeric, 3, 3>;
using _BA_Position3 = Eigen::Matrix<_BA_Numeric, 3, 1>;
using _BA_Pose = Eigen::Transform<_BA_Numeric, 3, 1>;
using _BA_PixelVec = std::vector<_BA_Pixel>;
using _BA_PointVec = std::vector<_BA_Point>;
using _BA_IndexVec = std::vector<_BA_Index>;

struct _BA_Camera {
  _BA_Camera() {}
  _BA_Camera(const _BA_Camera &camera)
      : fx(camera.fx), fy(camera.fy), cx(camera.cx), cy(camera.cy), pose_this_to_cam0(camera.pose_this_to_cam0) {}
  _BA_Camera &operator=(const _BA_Camera &camera) = default;
  _BA_Numeric fx{0.0};
  _BA_Numeric fy{0.0};
  _BA_Numeric cx{0.0};
  _BA_Numeric cy{0.0};
  _BA_Pose pose_this_to_cam0;  // body -> this camera
};

class FullBundleAdjustmentSolver {
 public:
  FullBundleAdjustmentSolver();
  ~FullBundleAdjustmentSolver();
  FullBundleAdjustmentSolver(const FullBundleAdjustmentSolver &) = delete;
  FullBundleAdjustmentSolver &operator=(const FullBundleAdjustmentSolver &) = delete;

  void Reset();

  void AddCamera(const _BA_Index camera_index, const _BA_Camera &camera);
  void AddPose(_BA_Pose *original_pose);
  void AddPoint(_BA_Point *original_point);
  void AddObservation(const _BA_Index index_camera, _BA_Pose *related_pose, _BA_Point *related_point,
                      const _BA_Pixel &pixel);

  void MakePoseFixed(_BA_Pose *original_pose_to_be_fixed);
  void MakePointFixed(_BA_Point *original_point_to_be_fixed);

  // Private in the reference (called inside Solve); public and idempotent
  // here, as the reference README lists it in the usage sequence.
  void FinalizeParameters();

  bool Solve(Options options, Summary *summary = nullptr);

  // (new) Read the CURRENT values behind every registered pose / point pointer again
  // and hand them to the finalized problem (ba_update_values): re-optimising the same
  // graph with new values costs no second FinalizeParameters.  The reference keeps its
  // own copies from AddPose / AddPoint across Solve calls (.cpp:44-70, :87-117) and
  // has no such call; without it this class, like the reference, continues from its
  // internal state.
  void ReloadParameterValues();

  std::string GetSolverStatistics() const;

  // GPU selection / console chatter (not in the reference)
  void SetDevice(int device_id) { device_id_ = device_id; }
  void SetVerbose(bool on) { verbose_ = on; }
  // Plain Gauss-Newton instead of Levenberg-Marquardt.  FullBundleAdjustmentSolver
  // itself never sets it (its Solve ignores options.solver_type, reference
  // :630-1044); FullBundleAdjustmentSolverRefactor does.
  void SetGaussNewton(bool on) { gauss_newton_ = on; }
  // Multi-GPU (new): this process owns landmark shard `rank` of
  // `world` (call before FinalizeParameters, with the SAME full problem
  // registered on every rank) and provides the sum-all-reduce the library calls
  // twice per LM iteration (ba_allreduce_fn of include/ba_hip.h; e.g.
  // multi_gpu::RcclAllReduce::Hook of utility/rccl_allreduce.h) and once more at
  // the end of Solve for the landmarks (ba_gather_points).  After Solve every rank
  // has written back all poses and ALL points (reference :1011-1022); OwnsPoint
  // tells which landmarks this rank linearised (all of them without a hook).
  void SetShard(int rank, int world) {
    shard_rank_ = rank;
    shard_world_ = world;
  }
  void SetAllReduce(int (*fn)(void *, int, void *, int64_t, void *), void *user);
  bool OwnsPoint(const _BA_Point *point) const;

 private:
  // stderr warnings about weakly connected poses / points (reference
  // core/full_bundle_adjustment_solver.cpp:310-341), called by Solve
  void CheckPoseAndPointConnectivity();

  struct Observation {
    int camera_index;
    int pose_index;
    int point_index;
    double u, v;
  };
  _BA_Numeric scaler_{0.01};
  _BA_Numeric inverse_scaler_{100.0};
  bool is_parameter_finalized_{false};
  bool verbose_{true};
  bool gauss_newton_{false};
  int device_id_{0};
  ba_handle *handle_{nullptr};
  int shard_rank_{0}, shard_world_{1};
  int (*allreduce_fn_)(void *, int, void *, int64_t, void *){nullptr};
  void *allreduce_user_{nullptr};
  std::vector<uint8_t> owned_points_;  // filled by Solve (all ones on a single GPU)

  std::vector<_BA_Index> camera_ids_;
  std::vector<_BA_Camera> cameras_;  // scaled copies
  std::unordered_map<_BA_Pose *, int> pose_index_;
  std::vector<_BA_Pose *> poses_;
  std::vector<_BA_Pose> T_jw_;  // inverted + scaled
  std::unordered_set<int> fixed_poses_;
  std::unordered_map<_BA_Point *, int> point_index_;
  std::vector<_BA_Point *> points_;
  std::vector<_BA_Point> X_;  // scaled
  std::unordered_set<int> fixed_points_;
  std::vector<Observation> observations_;

  _BA_Size_t num_fixed_poses_{0};
  _BA_Size_t num_fixed_points_{0};
};

}  // namespace analytic_solver
}  // namespace visual_navigation
#endif
