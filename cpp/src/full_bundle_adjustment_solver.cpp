// Host facade of the full BA solver: see the header.  Reference line numbers
// refer to core/full_bundle_adjustment_solver.cpp.
#include "core/full_bundle_adjustment_solver.h"

#include <algorithm>
#include <array>
#include <stdexcept>

#include "ba_hip.h"

namespace visual_navigation {
namespace analytic_solver {

namespace {
void Pack12(const _BA_Pose &T, double *out) {  // row-major R, then t
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[3 * r + c] = T.linear()(r, c);
  for (int r = 0; r < 3; ++r) out[9 + r] = T.translation()(r);
}
_BA_Pose Unpack12(const double *in) {
  _BA_Pose T;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T.linear()(r, c) = in[3 * r + c];
    T.translation()(r) = in[9 + r];
  }
  return T;
}
void Check(int rc, const char *what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + ba_last_error());
}
}  // namespace

FullBundleAdjustmentSolver::FullBundleAdjustmentSolver() {
  if (verbose_) std::cout << "SparseBundleAdjustmentSolver() - initialize.\n";  // :41
}

FullBundleAdjustmentSolver::~FullBundleAdjustmentSolver() {
  if (handle_) ba_destroy(handle_);
}

void FullBundleAdjustmentSolver::Reset() {  // :44-70
  if (handle_) ba_destroy(handle_);
  handle_ = nullptr;
  owned_points_.clear();
  is_parameter_finalized_ = false;
  camera_ids_.clear();
  cameras_.clear();
  pose_index_.clear();
  poses_.clear();
  T_jw_.clear();
  fixed_poses_.clear();
  point_index_.clear();
  points_.clear();
  X_.clear();
  fixed_points_.clear();
  observations_.clear();
  num_fixed_poses_ = num_fixed_points_ = 0;
}

void FullBundleAdjustmentSolver::AddCamera(const _BA_Index camera_index, const _BA_Camera &camera) {  // :72-85
  if (std::find(camera_ids_.begin(), camera_ids_.end(), camera_index) != camera_ids_.end())
    return;  // unordered_map::insert keeps the first
  _BA_Camera scaled = camera;
  scaled.pose_this_to_cam0.translation() *= scaler_;
  scaled.fx *= scaler_;
  scaled.fy *= scaler_;
  scaled.cx *= scaler_;
  scaled.cy *= scaler_;
  camera_ids_.push_back(camera_index);
  cameras_.push_back(scaled);
  if (verbose_)
    std::cout << "New camera is added.\n  fx: " << scaled.fx << ", fy: " << scaled.fy << ", cx: " << scaled.cx
              << ", cy: " << scaled.cy << "\n";
}

void FullBundleAdjustmentSolver::AddPose(_BA_Pose *original_pose) {  // :87-101
  if (is_parameter_finalized_) {
    std::cerr << TEXT_YELLOW("Cannot enroll parameter. (is_parameter_finalized_ == true)") << std::endl;
    return;
  }
  if (pose_index_.count(original_pose)) return;
  _BA_Pose T_jw = original_pose->inverse();
  T_jw.translation() = T_jw.translation() * scaler_;
  pose_index_[original_pose] = static_cast<int>(poses_.size());
  poses_.push_back(original_pose);
  T_jw_.push_back(T_jw);
}

void FullBundleAdjustmentSolver::AddPoint(_BA_Point *original_point) {  // :103-117
  if (is_parameter_finalized_) {
    std::cerr << TEXT_YELLOW("Cannot enroll parameter. (is_parameter_finalized_ == true)\n");
    return;
  }
  if (point_index_.count(original_point)) return;
  point_index_[original_point] = static_cast<int>(points_.size());
  points_.push_back(original_point);
  X_.push_back((*original_point) * scaler_);
}

void FullBundleAdjustmentSolver::MakePoseFixed(_BA_Pose *original_poseptr) {  // :119-134
  if (is_parameter_finalized_) {
    std::cerr << TEXT_YELLOW("Cannot enroll parameter. (is_parameter_finalized_ == true)\n");
    return;
  }
  if (original_poseptr == nullptr) {
    std::cerr << "Empty pointer is conveyed. Skip this one.\n";
    return;
  }
  auto it = pose_index_.find(original_poseptr);
  if (it == pose_index_.end()) throw std::runtime_error("There is no pointer in the BA pose pool.");
  fixed_poses_.insert(it->second);
  ++num_fixed_poses_;
}

void FullBundleAdjustmentSolver::MakePointFixed(_BA_Point *original_pointptr_to_be_fixed) {  // :136-153
  if (is_parameter_finalized_) {
    std::cerr << TEXT_YELLOW("Cannot enroll parameter. (is_parameter_finalized_ == true)\n");
    return;
  }
  if (original_pointptr_to_be_fixed == nullptr) {
    std::cerr << "Empty pointer is conveyed. Skip this one.\n";
    return;
  }
  auto it = point_index_.find(original_pointptr_to_be_fixed);
  if (it == point_index_.end()) throw std::runtime_error("There is no pointer in the BA point pool.");
  fixed_points_.insert(it->second);
  ++num_fixed_points_;
}

void FullBundleAdjustmentSolver::AddObservation(const _BA_Index camera_index, _BA_Pose *related_pose,
                                                _BA_Point *related_point, const _BA_Pixel &pixel) {  // :155-180
  const auto cam_it = std::find(camera_ids_.begin(), camera_ids_.end(), camera_index);
  if (cam_it == camera_ids_.end()) {
    std::cerr << TEXT_RED("Invalid camera index.\n");
    return;
  }
  const auto pose_it = pose_index_.find(related_pose);
  if (pose_it == pose_index_.end()) {
    std::cerr << TEXT_RED("Nonexisting pose.\n");
    return;
  }
  const auto point_it = point_index_.find(related_point);
  if (point_it == point_index_.end()) {
    std::cerr << TEXT_RED("Nonexisting point.\n");
    return;
  }
  Observation o;
  o.camera_index = static_cast<int>(cam_it - camera_ids_.begin());
  o.pose_index = pose_it->second;
  o.point_index = point_it->second;
  o.u = pixel(0) * scaler_;
  o.v = pixel(1) * scaler_;
  observations_.push_back(o);
}

void FullBundleAdjustmentSolver::FinalizeParameters() {  // :182-206, :243-308, :668-700
  if (is_parameter_finalized_) return;
  if (cameras_.empty() || poses_.empty() || points_.empty())
    throw std::runtime_error("FinalizeParameters: cameras, poses and points must be added first");
  Check(ba_create(&handle_, device_id_), "ba_create");
  std::vector<double> intr, T_cj(12 * cameras_.size());
  for (size_t c = 0; c < cameras_.size(); ++c) {
    intr.insert(intr.end(), {cameras_[c].fx, cameras_[c].fy, cameras_[c].cx, cameras_[c].cy});
    Pack12(cameras_[c].pose_this_to_cam0, &T_cj[12 * c]);
  }
  Check(ba_set_cameras(handle_, static_cast<int>(cameras_.size()), intr.data(), T_cj.data()), "ba_set_cameras");
  std::vector<double> T(12 * poses_.size()), X(3 * points_.size());
  std::vector<uint8_t> pose_fixed(poses_.size(), 0), point_fixed(points_.size(), 0);
  for (size_t p = 0; p < poses_.size(); ++p) {
    Pack12(T_jw_[p], &T[12 * p]);
    pose_fixed[p] = fixed_poses_.count(static_cast<int>(p)) > 0;
  }
  for (size_t q = 0; q < points_.size(); ++q) {
    for (int r = 0; r < 3; ++r) X[3 * q + r] = X_[q](r);
    point_fixed[q] = fixed_points_.count(static_cast<int>(q)) > 0;
  }
  Check(ba_set_poses(handle_, static_cast<int>(poses_.size()), T.data(), pose_fixed.data()), "ba_set_poses");
  Check(ba_set_points(handle_, static_cast<int>(points_.size()), X.data(), point_fixed.data()), "ba_set_points");
  std::vector<int32_t> oc(observations_.size()), op(observations_.size()), oq(observations_.size());
  std::vector<double> uv(2 * observations_.size());
  for (size_t k = 0; k < observations_.size(); ++k) {  // insertion order matters (:826)
    oc[k] = observations_[k].camera_index;
    op[k] = observations_[k].pose_index;
    oq[k] = observations_[k].point_index;
    uv[2 * k] = observations_[k].u;
    uv[2 * k + 1] = observations_[k].v;
  }
  Check(ba_set_observations(handle_, static_cast<int64_t>(oc.size()), oc.data(), op.data(), oq.data(), uv.data()),
        "ba_set_observations");
  if (shard_world_ > 1) Check(ba_set_shard(handle_, shard_rank_, shard_world_), "ba_set_shard");
  Check(ba_finalize(handle_), "ba_finalize");
  if (allreduce_fn_) Check(ba_set_allreduce(handle_, allreduce_fn_, allreduce_user_), "ba_set_allreduce");
  is_parameter_finalized_ = true;
}

void FullBundleAdjustmentSolver::ReloadParameterValues() {
  if (!is_parameter_finalized_) {  // nothing planned yet: refresh the copies FinalizeParameters will read
    for (size_t p = 0; p < poses_.size(); ++p) {
      T_jw_[p] = poses_[p]->inverse();
      T_jw_[p].translation() = T_jw_[p].translation() * scaler_;
    }
    for (size_t q = 0; q < points_.size(); ++q) X_[q] = (*points_[q]) * scaler_;
    return;
  }
  std::vector<double> T(12 * poses_.size()), X(3 * points_.size());
  for (size_t p = 0; p < poses_.size(); ++p) {
    T_jw_[p] = poses_[p]->inverse();
    T_jw_[p].translation() = T_jw_[p].translation() * scaler_;
    Pack12(T_jw_[p], &T[12 * p]);
  }
  for (size_t q = 0; q < points_.size(); ++q) {
    X_[q] = (*points_[q]) * scaler_;
    for (int r = 0; r < 3; ++r) X[3 * q + r] = X_[q](r);
  }
  Check(ba_update_values(handle_, T.data(), X.data()), "ba_update_values");
}

void FullBundleAdjustmentSolver::SetAllReduce(int (*fn)(void *, int, void *, int64_t, void *), void *user) {
  allreduce_fn_ = fn;
  allreduce_user_ = user;
  if (is_parameter_finalized_) Check(ba_set_allreduce(handle_, fn, user), "ba_set_allreduce");
}

bool FullBundleAdjustmentSolver::OwnsPoint(const _BA_Point *point) const {
  const auto it = point_index_.find(const_cast<_BA_Point *>(point));
  if (it == point_index_.end()) return false;
  return owned_points_.empty() || owned_points_[static_cast<size_t>(it->second)] != 0;
}

std::string FullBundleAdjustmentSolver::GetSolverStatistics() const {  // :208-239 (returns "")
  const size_t n_opt_pose = poses_.size() - fixed_poses_.size();
  const size_t n_opt_point = points_.size() - fixed_points_.size();
  std::cout << "| Bundle Adjustment Statistics:\n"
            << "| # cameras in rigid body system: " << cameras_.size() << "\n"
            << "|   " << TEXT_CYAN("(Note: The reference camera is 'camera_list_[0]'.)") << "\n"
            << "|             # of total poses: " << poses_.size() << "\n"
            << "|               - # fix  poses: " << num_fixed_poses_ << "\n"
            << "|               - # opt. poses: " << n_opt_pose << "\n"
            << "|            # of total points: " << points_.size() << "\n"
            << "|              - # fix  points: " << num_fixed_points_ << "\n"
            << "|              - # opt. points: " << n_opt_point << "\n"
            << "|            # of observations: " << observations_.size() << "\n"
            << "|                Jacobian size: " << 6 * observations_.size() << " rows x "
            << 3 * n_opt_point + 6 * n_opt_pose << " cols\n"
            << "|                Residual size: " << 2 * observations_.size() << " rows\n"
            << std::endl;
  return std::string();
}

void FullBundleAdjustmentSolver::CheckPoseAndPointConnectivity() {  // :310-341
  static constexpr int kMinNumObservedPoints = 5;
  static constexpr int kMinNumRelatedPoses = 2;
  // distinct related points per pose (only "fewer than 5" matters: the first
  // five distinct ids are kept) and distinct related poses per point (first id
  // + "a second one exists"); fixed partners count (:684-693)
  std::vector<std::array<int, kMinNumObservedPoints>> seen(poses_.size());
  std::vector<int> n_seen(poses_.size(), 0), first_pose(points_.size(), -1);
  std::vector<char> second_pose(points_.size(), 0);
  for (const Observation &o : observations_) {
    int &n = n_seen[o.pose_index];
    if (n < kMinNumObservedPoints) {
      auto &ids = seen[o.pose_index];
      if (std::find(ids.begin(), ids.begin() + n, o.point_index) == ids.begin() + n) ids[n++] = o.point_index;
    }
    if (first_pose[o.point_index] < 0)
      first_pose[o.point_index] = o.pose_index;
    else if (first_pose[o.point_index] != o.pose_index)
      second_pose[o.point_index] = 1;
  }
  // optimisation indices: input order of the non-fixed entries (the reference's
  // are unordered_map iteration order)
  int j_opt = 0;
  for (size_t p = 0; p < poses_.size(); ++p) {
    if (fixed_poses_.count(static_cast<int>(p))) continue;
    if (n_seen[p] < kMinNumObservedPoints)
      std::cerr << TEXT_YELLOW(std::to_string(j_opt) +
                               "-th pose: It might diverge because some frames "
                               "have insufficient related points.")
                << std::endl;
    ++j_opt;
  }
  int i_opt = 0;
  for (size_t q = 0; q < points_.size(); ++q) {
    if (fixed_points_.count(static_cast<int>(q))) continue;
    const int num_related_pose = (first_pose[q] >= 0 ? 1 : 0) + (second_pose[q] ? 1 : 0);
    if (num_related_pose < kMinNumRelatedPoses)
      std::cerr << TEXT_YELLOW(std::to_string(i_opt) +
                               "-th point: It might diverge because some "
                               "points have insufficient related poses.")
                << std::endl;
    ++i_opt;
  }
}

bool FullBundleAdjustmentSolver::Solve(Options options, Summary *summary) {  // :630-1044
  timer::StopWatch stopwatch("BundleAdjustmentSolver::Solve");
  stopwatch.Start();
  if (summary != nullptr) {
    summary->max_iteration_ = options.iteration_handle.max_num_iterations;
    summary->threshold_cost_change_ = options.convergence_handle.threshold_cost_change;
    summary->threshold_step_size_ = options.convergence_handle.threshold_step_size;
    summary->convergence_status_ = true;
  }
  FinalizeParameters();
  if (verbose_) GetSolverStatistics();
  CheckPoseAndPointConnectivity();  // :703

  ba_options o;
  o.threshold_step_size = options.convergence_handle.threshold_step_size;
  o.threshold_cost_change = options.convergence_handle.threshold_cost_change;
  o.threshold_huber_loss = options.outlier_handle.threshold_huber_loss;
  o.threshold_outlier_rejection = options.outlier_handle.threshold_outlier_rejection;
  o.max_num_iterations = options.iteration_handle.max_num_iterations;
  o.initial_lambda = options.trust_region_handle.initial_lambda;
  o.decrease_ratio_lambda = options.trust_region_handle.decrease_ratio_lambda;
  o.increase_ratio_lambda = options.trust_region_handle.increase_ratio_lambda;
  o.gauss_newton = gauss_newton_ ? 1 : 0;
  std::vector<ba_iter_info> rows(static_cast<size_t>(std::max(1, o.max_num_iterations)));
  int n_iter = 0, converged = 0;
  Check(ba_solve(handle_, &o, rows.data(), static_cast<int>(rows.size()), &n_iter, &converged), "ba_solve");

  // write back through the caller's pointers (:1011-1022)
  std::vector<double> T(12 * poses_.size()), X(3 * points_.size());
  Check(ba_get_poses(handle_, T.data()), "ba_get_poses");
  // sharded: one final sum-all-reduce of the owned landmark rows, so that EVERY rank
  // writes back EVERY point, as the reference's contract says (:1018-1022)
  owned_points_.assign(points_.size(), 1);
  Check(ba_get_points(handle_, X.data(), owned_points_.data()), "ba_get_points");  // (+ which landmarks are this shard's)
  std::vector<uint8_t> valid(owned_points_);
  if (shard_world_ > 1 && allreduce_fn_) {
    Check(ba_gather_points(handle_), "ba_gather_points");
    Check(ba_get_points(handle_, X.data(), valid.data()), "ba_get_points");
  }
  for (size_t p = 0; p < poses_.size(); ++p) {
    if (fixed_poses_.count(static_cast<int>(p))) continue;
    _BA_Pose T_jw = Unpack12(&T[12 * p]);
    T_jw_[p] = T_jw;
    T_jw.translation() *= inverse_scaler_;
    *poses_[p] = T_jw.inverse();
  }
  for (size_t q = 0; q < points_.size(); ++q) {
    if (fixed_points_.count(static_cast<int>(q)) || !valid[q]) continue;
    X_[q] = _BA_Point(X[3 * q], X[3 * q + 1], X[3 * q + 2]);
    *points_[q] = X_[q] * inverse_scaler_;
  }
  if (summary != nullptr) {
    for (int k = 0; k < n_iter && k < static_cast<int>(rows.size()); ++k) {
      OptimizationInfo info;
      info.cost = rows[k].cost;
      info.cost_change = rows[k].cost_change;
      info.average_reprojection_error = rows[k].average_reprojection_error;
      info.abs_step = rows[k].abs_step;
      info.abs_gradient = 0;
      info.damping_term = rows[k].damping_term;
      info.iter_time = rows[k].iter_time_ms;
      info.iteration_status = static_cast<IterationStatus>(rows[k].iteration_status);
      summary->optimization_info_list_.push_back(info);
    }
    summary->convergence_status_ = converged != 0;
    summary->total_time_in_millisecond_ = stopwatch.GetLapTimeFromStart();
  }
  return true;  // the reference always returns true (:1043)
}

}  // namespace analytic_solver
}  // namespace visual_navigation
