// GPU test of the C++ host facade: the stereo wall scene of reference
// test/test_ba.cpp (seeded), solved through
// visual_navigation::analytic_solver::FullBundleAdjustmentSolver with the
// calls a reference user makes, and checked against the CPU oracle run on the
// same registered problem (trajectory, final poses, final points).  Also the
// pose-only facade against the oracle.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "core/full_bundle_adjustment_solver.h"
#include "utility/bal_io.h"
#include "core/full_bundle_adjustment_solver_refactor.h"
#include "core/pose_only_bundle_adjustment_solver.h"
#include "eigen3/Eigen/Dense"
#include "eigen3/Eigen/Geometry"
#include "../../oracle/ba_oracle.h"

using namespace visual_navigation::analytic_solver;
using Pose = Eigen::Transform<double, 3, 1>;
using Point = Eigen::Matrix<double, 3, 1>;

static int g_fail = 0;
#define EXPECT(cond, ...)                                  \
  do {                                                     \
    if (!(cond)) {                                         \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__);     \
      std::printf(__VA_ARGS__);                            \
      std::printf("\n");                                   \
      ++g_fail;                                            \
    }                                                      \
  } while (0)

static void Pack12(const Pose &T, double *o) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o[3 * r + c] = T.linear()(r, c);
    o[9 + r] = T.translation()(r);
  }
}

int main() {
  // ---------------- scene (law of reference test/test_ba.cpp:53-232) ----------------
  const int num_total_poses = 60, num_fixed_poses = 5;
  std::mt19937 gen(20240601u);
  std::uniform_real_distribution<float> point_err(-0.5f, 0.5f), position_err(-0.1f, 0.1f);
  _BA_Camera cam_left, cam_right;
  cam_left.fx = cam_right.fx = 525.0;
  cam_left.fy = cam_right.fy = 525.0;
  cam_left.cx = cam_right.cx = 320.0;
  cam_left.cy = cam_right.cy = 240.0;
  cam_left.pose_this_to_cam0 = Pose::Identity();
  Pose left_to_right = Pose::Identity();
  left_to_right.translation().x() += 0.12;
  cam_right.pose_this_to_cam0 = left_to_right.inverse();
  std::vector<_BA_Camera> cameras{cam_left, cam_right};

  std::vector<Point> true_points;
  for (float z = 1.7f; z <= 5.7f; z += 0.4f)
    for (float y = 0.0f; y <= 26.0f; y += 0.4f) true_points.push_back(Point(8.5, y, z));
  Pose base_to_camera = Pose::Identity();
  base_to_camera.linear() = Eigen::AngleAxis<double>(M_PI_2, Point::UnitY()).toRotationMatrix() *
                            Eigen::AngleAxis<double>(-M_PI_2, Point::UnitZ()).toRotationMatrix();
  Pose world_to_base = Pose::Identity();
  world_to_base.linear() = Eigen::AngleAxis<double>(-0.1, Point::UnitZ()).toRotationMatrix();
  world_to_base.translation() = Point(-4.0, -2.5, 0.0);
  std::vector<Pose> true_poses;
  std::unordered_map<int, Pose> pose_pool;         // node-stable storage, like the reference test
  std::unordered_map<int, Point> point_pool;
  for (int k = 0; k < num_total_poses; ++k) {
    world_to_base.linear() =
        world_to_base.linear() * Eigen::AngleAxis<double>(0.005f, Point::UnitZ()).toRotationMatrix();
    world_to_base.translation().x() += 0.005f;
    world_to_base.translation().y() += 0.2f;
    true_poses.push_back(world_to_base * base_to_camera);
    pose_pool[k] = true_poses.back();
  }
  for (int k = num_fixed_poses; k < num_total_poses; ++k)
    for (int a = 0; a < 3; ++a) pose_pool[k].translation()(a) += position_err(gen);
  for (size_t i = 0; i < true_points.size(); ++i) {
    point_pool[(int)i] = true_points[i];
    for (int a = 0; a < 3; ++a) point_pool[(int)i](a) += point_err(gen);
  }
  struct Obs { int cam, pose, point; _BA_Pixel px; };
  std::vector<Obs> obs;
  for (int j = 0; j < num_total_poses; ++j) {
    const Pose cam_from_world = true_poses[j].inverse();
    for (int c = 0; c < 2; ++c)
      for (size_t i = 0; i < true_points.size(); ++i) {
        const Point local = cameras[c].pose_this_to_cam0 * (cam_from_world * true_points[i]);
        const float inv_z = 1.0 / local(2);
        _BA_Pixel px(cameras[c].fx * local(0) * inv_z + cameras[c].cx,
                     cameras[c].fy * local(1) * inv_z + cameras[c].cy);
        if (px.x() < 640 && px.x() > 0 && px.y() < 480 && px.y() > 0) obs.push_back({c, j, (int)i, px});
      }
  }
  std::printf("scene: %d poses, %zu landmarks, %zu observations\n", num_total_poses, true_points.size(), obs.size());
  EXPECT(obs.size() == 34019, "observation count %zu", obs.size());

  // ---------------- oracle inputs (the facade's host preprocessing, restated) ----------------
  const double s = 0.01;
  std::vector<double> cam_intr, cam_T(24), pose_T(12 * num_total_poses), X(3 * true_points.size()), uv;
  std::vector<uint8_t> pose_fixed(num_total_poses, 0), pt_fixed(true_points.size(), 0);
  std::vector<int32_t> oc, op, oq;
  for (int c = 0; c < 2; ++c) {
    cam_intr.insert(cam_intr.end(), {cameras[c].fx * s, cameras[c].fy * s, cameras[c].cx * s, cameras[c].cy * s});
    Pose T = cameras[c].pose_this_to_cam0;
    T.translation() *= s;
    Pack12(T, &cam_T[12 * c]);
  }
  for (int j = 0; j < num_total_poses; ++j) {
    Pose T = pose_pool[j].inverse();
    T.translation() = T.translation() * s;
    Pack12(T, &pose_T[12 * j]);
    pose_fixed[j] = j < num_fixed_poses;
  }
  for (size_t i = 0; i < true_points.size(); ++i)
    for (int a = 0; a < 3; ++a) X[3 * i + a] = point_pool[(int)i](a) * s;
  for (const Obs &o : obs) {
    oc.push_back(o.cam); op.push_back(o.pose); oq.push_back(o.point);
    uv.push_back(o.px.x() * s); uv.push_back(o.px.y() * s);
  }
  ba_oracle *orc = ba_oracle_create(2, cam_intr.data(), cam_T.data(), num_total_poses, pose_T.data(),
                                    pose_fixed.data(), (int)true_points.size(), X.data(), pt_fixed.data(),
                                    (int64_t)oc.size(), oc.data(), op.data(), oq.data(), uv.data());

  // ---------------- the calls a reference user makes (test_ba.cpp:235-297) ----------------
  FullBundleAdjustmentSolver ba_solver;
  ba_solver.SetVerbose(false);
  for (int c = 0; c < 2; ++c) ba_solver.AddCamera(c, cameras[c]);
  for (int j = 0; j < num_total_poses; ++j) ba_solver.AddPose(&pose_pool[j]);
  for (size_t i = 0; i < true_points.size(); ++i) ba_solver.AddPoint(&point_pool[(int)i]);
  for (int j = 0; j < num_fixed_poses; ++j) ba_solver.MakePoseFixed(&pose_pool[j]);
  ba_solver.MakePointFixed({});  // nullptr: message + skip
  bool threw = false;
  Pose stranger;
  try { ba_solver.MakePoseFixed(&stranger); } catch (const std::runtime_error &) { threw = true; }
  EXPECT(threw, "unknown pose pointer must throw");
  for (const Obs &o : obs) ba_solver.AddObservation(o.cam, &pose_pool[o.pose], &point_pool[o.point], o.px);
  ba_solver.AddObservation(7, &pose_pool[0], &point_pool[0], obs[0].px);  // invalid camera: dropped

  Options options;
  options.iteration_handle.max_num_iterations = 40;
  options.convergence_handle.threshold_cost_change = 1e-6f;
  options.convergence_handle.threshold_step_size = 1e-6f;
  Summary summary;
  // CheckPoseAndPointConnectivity (reference :310-341, called at :703): the wall
  // scene has landmarks no camera sees -> one "-th point" warning each on stderr,
  // and every pose sees far more than 5 points -> no "-th pose" warning
  std::stringstream captured;
  std::streambuf *old_err = std::cerr.rdbuf(captured.rdbuf());
  const bool solved = ba_solver.Solve(options, &summary);
  std::cerr.rdbuf(old_err);
  EXPECT(solved, "Solve returns true");
  {
    std::vector<char> seen_pt(true_points.size(), 0);
    std::vector<int> poses_of_pt(true_points.size(), -1);
    size_t weak_points = 0;
    for (const Obs &o : obs) {
      if (poses_of_pt[o.point] == -1) poses_of_pt[o.point] = o.pose;
      else if (poses_of_pt[o.point] != o.pose) seen_pt[o.point] = 1;
    }
    for (size_t i = 0; i < true_points.size(); ++i) weak_points += !seen_pt[i];
    const std::string text = captured.str();
    size_t n_pt_warn = 0, n_pose_warn = 0, at = 0;
    while ((at = text.find("-th point: It might diverge because some points have insufficient related poses.", at)) !=
           std::string::npos) { ++n_pt_warn; ++at; }
    at = 0;
    while ((at = text.find("-th pose: It might diverge", at)) != std::string::npos) { ++n_pose_warn; ++at; }
    EXPECT(weak_points > 0 && n_pt_warn == weak_points, "connectivity warnings: %zu points warned, %zu weak", n_pt_warn,
           weak_points);
    EXPECT(n_pose_warn == 0, "unexpected pose connectivity warnings: %zu", n_pose_warn);
    std::printf("connectivity warnings: %zu weakly observed points reported on stderr\n", n_pt_warn);
  }
  std::printf("%s\n", summary.BriefReport().c_str());

  ba_oracle_options oo{1e-6f, 1e-6f, 1.0f, 2.0f, 40, 100.0f, 0.33f, 3.0f};
  std::vector<ba_oracle_iter> orows(40);
  int oconv = 0;
  const int on = ba_oracle_solve(orc, &oo, orows.data(), 40, &oconv);
  const auto &rows = summary.GetOptimizationInfoList();
  EXPECT((int)rows.size() == on, "iterations %zu vs oracle %d", rows.size(), on);
  EXPECT(summary.IsConverged() == (oconv != 0), "convergence flag");
  for (size_t k = 0; k < rows.size() && (int)k < on; ++k) {
    EXPECT((int)rows[k].iteration_status == orows[k].iteration_status, "status at %zu", k);
    EXPECT(std::fabs(rows[k].cost - orows[k].cost) <= 1e-7 * std::fabs(orows[k].cost), "cost at %zu: %.12e vs %.12e", k,
           rows[k].cost, orows[k].cost);
    EXPECT(std::fabs(rows[k].damping_term - orows[k].damping_term) <= 1e-12 * orows[k].damping_term, "lambda at %zu", k);
  }
  // final parameters: user objects were written back in USER units
  std::vector<double> oT(12 * num_total_poses), oX(3 * true_points.size());
  ba_oracle_get_poses(orc, oT.data());
  ba_oracle_get_points(orc, oX.data());
  double max_dp = 0, max_dx = 0;
  for (int j = 0; j < num_total_poses; ++j) {
    Pose T_jw;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T_jw.linear()(r, c) = oT[12 * j + 3 * r + c];
      T_jw.translation()(r) = oT[12 * j + 9 + r] / s;
    }
    const Pose expect = T_jw.inverse();
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) max_dp = std::fmax(max_dp, std::fabs(expect.linear()(r, c) - pose_pool[j].linear()(r, c)));
      max_dp = std::fmax(max_dp, std::fabs(expect.translation()(r) - pose_pool[j].translation()(r)));
    }
  }
  for (size_t i = 0; i < true_points.size(); ++i)
    for (int a = 0; a < 3; ++a) max_dx = std::fmax(max_dx, std::fabs(oX[3 * i + a] / s - point_pool[(int)i](a)));
  std::printf("max |pose - oracle| = %.3e   max |point - oracle| = %.3e (user units)\n", max_dp, max_dx);
  EXPECT(max_dp < 1e-6 && max_dx < 1e-6, "final parameters differ from the oracle");
  ba_oracle_destroy(orc);

  // ---------------- re-solve with new VALUES, no re-planning (ReloadParameterValues) ----------------
  // the caller puts the initial values back into its own objects and has the solver read
  // them again: the second Solve repeats the first trajectory bit for bit; without the
  // reload a further Solve continues from the internal state, as the reference does
  {
    std::mt19937 gen2(20240601u);
    std::uniform_real_distribution<float> pe(-0.5f, 0.5f), se(-0.1f, 0.1f);
    for (int k = 0; k < num_total_poses; ++k) pose_pool[k] = true_poses[k];
    for (int k = num_fixed_poses; k < num_total_poses; ++k)
      for (int a = 0; a < 3; ++a) pose_pool[k].translation()(a) += se(gen2);
    for (size_t i = 0; i < true_points.size(); ++i) {
      point_pool[(int)i] = true_points[i];
      for (int a = 0; a < 3; ++a) point_pool[(int)i](a) += pe(gen2);
    }
    ba_solver.ReloadParameterValues();
    Summary again;
    std::streambuf *keep = std::cerr.rdbuf(captured.rdbuf());
    ba_solver.Solve(options, &again);
    std::cerr.rdbuf(keep);
    const auto &r2 = again.GetOptimizationInfoList();
    EXPECT(r2.size() == rows.size(), "re-solve: %zu iterations vs %zu", r2.size(), rows.size());
    bool same = r2.size() == rows.size();
    for (size_t k = 0; same && k < rows.size(); ++k)
      same = r2[k].cost == rows[k].cost && r2[k].damping_term == rows[k].damping_term &&
             r2[k].iteration_status == rows[k].iteration_status;
    EXPECT(same, "re-solve after ReloadParameterValues must repeat the first trajectory bit for bit");
    Summary cont;
    keep = std::cerr.rdbuf(captured.rdbuf());
    ba_solver.Solve(options, &cont);
    std::cerr.rdbuf(keep);
    EXPECT(!cont.GetOptimizationInfoList().empty() &&
               cont.GetOptimizationInfoList().front().cost < 0.5 * rows.front().cost,
           "a Solve without reload continues from the internal state");
    std::printf("re-solve: ReloadParameterValues repeats the first trajectory (%zu iterations), a further Solve continues\n",
                r2.size());
  }

  // ---------------- refactored API, Gauss-Newton mode (reference test_ba_refactor.cpp:236-299) ----------------
  {
    std::unordered_map<int, Pose> poses2;
    std::unordered_map<int, Point> points2;
    for (int j = 0; j < num_total_poses; ++j) poses2[j] = true_poses[j];
    for (int j = num_fixed_poses; j < num_total_poses; ++j) poses2[j].translation().x() += 0.01 * ((j % 7) - 3);
    for (size_t i = 0; i < true_points.size(); ++i) {
      points2[(int)i] = true_points[i];
      points2[(int)i].z() += 0.02 * ((int)(i % 5) - 2);
    }
    // oracle on the same registered problem
    std::vector<double> pT(12 * num_total_poses), pX(3 * true_points.size());
    for (int j = 0; j < num_total_poses; ++j) {
      Pose T = poses2[j].inverse();
      T.translation() = T.translation() * s;
      Pack12(T, &pT[12 * j]);
    }
    for (size_t i = 0; i < true_points.size(); ++i)
      for (int a = 0; a < 3; ++a) pX[3 * i + a] = points2[(int)i](a) * s;
    ba_oracle *o2 = ba_oracle_create(2, cam_intr.data(), cam_T.data(), num_total_poses, pT.data(), pose_fixed.data(),
                                     (int)true_points.size(), pX.data(), pt_fixed.data(), (int64_t)oc.size(),
                                     oc.data(), op.data(), oq.data(), uv.data());
    FullBundleAdjustmentSolverRefactor rf;
    rf.SetVerbose(false);
    for (int c = 0; c < 2; ++c) {
      OptimizerCamera oc2;
      oc2.fx = cameras[c].fx; oc2.fy = cameras[c].fy; oc2.cx = cameras[c].cx; oc2.cy = cameras[c].cy;
      oc2.camera_to_body_pose = cameras[c].pose_this_to_cam0;
      rf.RegisterCamera(c, oc2);
    }
    for (int j = 0; j < num_total_poses; ++j) rf.RegisterWorldToBodyPose(&poses2[j]);
    for (size_t i = 0; i < true_points.size(); ++i) rf.RegisterWorldPoint(&points2[(int)i]);
    for (int j = 0; j < num_fixed_poses; ++j) rf.MakePoseFixed(&poses2[j]);
    for (const Obs &ob : obs) rf.AddObservation(ob.cam, &poses2[ob.pose], &points2[ob.point], ob.px);
    Options gopt;
    gopt.solver_type = SolverType::GAUSS_NEWTON;
    gopt.iteration_handle.max_num_iterations = 6;
    gopt.convergence_handle.threshold_cost_change = 0.f;
    gopt.convergence_handle.threshold_step_size = 0.f;
    gopt.trust_region_handle.initial_lambda = 1e-3f;
    Summary gs;
    EXPECT(rf.Solve(gopt, &gs), "refactor Solve");
    ba_oracle_options go{0.f, 0.f, 1.0f, 2.0f, 6, 1e-3f, 0.33f, 3.0f, 1};
    std::vector<ba_oracle_iter> grows(6);
    int gconv = 0;
    const int gn = ba_oracle_solve(o2, &go, grows.data(), 6, &gconv);
    const auto &gr = gs.GetOptimizationInfoList();
    EXPECT((int)gr.size() == gn && gn == 6, "GN iterations %zu vs %d", gr.size(), gn);
    for (size_t k = 0; k < gr.size() && (int)k < gn; ++k) {
      EXPECT((int)gr[k].iteration_status == 0 && grows[k].iteration_status == 0, "GN status at %zu", k);
      EXPECT(std::fabs(gr[k].cost - grows[k].cost) <= 1e-7 * std::fabs(grows[k].cost) + 1e-12, "GN cost at %zu", k);
    }
    std::printf("refactor API, Gauss-Newton: cost %.4e -> %.4e in %d iterations (oracle %.4e)\n", gr.front().cost,
                gr.back().cost, gn, grows[gn - 1].cost);
    bool threw2 = false;
    gopt.solver_type = SolverType::UNDEFINED;
    try { rf.Solve(gopt, nullptr); } catch (const std::runtime_error &) { threw2 = true; }
    EXPECT(threw2, "undefined solver_type must throw");
    ba_oracle_destroy(o2);
  }

  // ---------------- pose-only facade ----------------
  {
    std::mt19937 g2(7u);
    std::uniform_real_distribution<float> dx(-1.7f, 1.7f), dy(-1.3f, 1.3f), dz(0.f, 5.f);
    Eigen::Isometry3f T_true = Eigen::Isometry3f::Identity();
    T_true.linear() = Eigen::AngleAxisf(-0.5f, Eigen::Vector3f::UnitY()).toRotationMatrix();
    T_true.translation() = Eigen::Vector3f(0.2f, 0.3f, -1.9f);
    std::vector<Eigen::Vector3f> Xw;
    std::vector<Eigen::Vector2f> px;
    const Eigen::Isometry3f Ti = T_true.inverse();
    for (int k = 0; k < 10000; ++k) {
      const Eigen::Vector3f w(dx(g2), dy(g2), dz(g2) + 1.2f);
      const Eigen::Vector3f l = Ti * w;
      Xw.push_back(w);
      px.push_back(Eigen::Vector2f(338.f * l(0) / l(2) + 320.f, 338.f * l(1) / l(2) + 240.f));
    }
    PoseOnlyBundleAdjustmentSolver po;
    Eigen::Isometry3f pose = Eigen::Isometry3f::Identity();
    std::vector<bool> mask;
    Options popt;
    popt.iteration_handle.max_num_iterations = 100;
    popt.convergence_handle.threshold_cost_change = 1e-6f;
    popt.convergence_handle.threshold_step_size = 1e-6f;
    popt.outlier_handle.threshold_outlier_rejection = 2.5f;
    Summary ps;
    EXPECT(po.Solve_Monocular_6Dof(Xw, px, 338.f, 338.f, 320.f, 240.f, pose, mask, popt, &ps), "pose-only solve");
    float err = 0.f;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) err = std::fmax(err, std::fabs(pose.linear()(r, c) - T_true.linear()(r, c)));
      err = std::fmax(err, std::fabs(pose.translation()(r) - T_true.translation()(r)));
    }
    std::printf("pose-only: %zu summary rows, %zu debug poses, max |T - T_true| = %.2e\n",
                ps.GetOptimizationInfoList().size(), po.GetDebugPoses().size(), err);
    EXPECT(err < 1e-3f, "pose-only did not recover the true pose");
    EXPECT(mask.size() == 10000, "mask size");
    // stereo entry point: right camera 0.12 m to the right, every third point unmatched
    Eigen::Isometry3f T_lr = Eigen::Isometry3f::Identity();
    T_lr.translation() = Eigen::Vector3f(0.12f, 0.f, 0.f);
    const Eigen::Isometry3f T_rl = T_lr.inverse();
    std::vector<Eigen::Vector2f> pxr;
    for (int k = 0; k < 10000; ++k) {
      const Eigen::Vector3f lr = T_rl * (Ti * Xw[k]);
      if (k % 3 == 0) pxr.push_back(Eigen::Vector2f(-1.f, -1.f));
      else pxr.push_back(Eigen::Vector2f(338.f * lr(0) / lr(2) + 320.f, 338.f * lr(1) / lr(2) + 240.f));
    }
    Eigen::Isometry3f pose_s = Eigen::Isometry3f::Identity();
    std::vector<bool> ml, mr;
    Summary ss;
    EXPECT(po.Solve_Stereo_6Dof(Xw, px, pxr, 338.f, 338.f, 320.f, 240.f, 338.f, 338.f, 320.f, 240.f, T_lr, pose_s,
                                ml, mr, popt, &ss),
           "stereo pose-only solve");
    float errs = 0.f;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) errs = std::fmax(errs, std::fabs(pose_s.linear()(r, c) - T_true.linear()(r, c)));
      errs = std::fmax(errs, std::fabs(pose_s.translation()(r) - T_true.translation()(r)));
    }
    std::printf("stereo pose-only: %zu summary rows, max |T - T_true| = %.2e\n",
                ss.GetOptimizationInfoList().size(), errs);
    EXPECT(errs < 1e-3f, "stereo pose-only did not recover the true pose");
    EXPECT(ml.size() == 10000 && mr.size() == 10000, "stereo mask sizes");
  }
  // ---------------- BAL scene file through the facade ----------------
  {
    const char *env = std::getenv("BA_TEST_BAL");
    const std::string path = env ? env : "tests/golden/bal_small.txt";
    visual_navigation::scene_io::BalProblem bal;
    std::string err;
    EXPECT(visual_navigation::scene_io::LoadBal(path, &bal, &err), "LoadBal(%s): %s", path.c_str(), err.c_str());
    EXPECT(!visual_navigation::scene_io::LoadBal(path, &bal, &err, /*undistort=*/false),
           "a distorted file must be refused without undistortion");
    EXPECT(visual_navigation::scene_io::LoadBal(path, &bal, &err), "LoadBal again");
    if (!bal.poses.empty()) {
      EXPECT(bal.poses.size() == 12 && bal.points.size() == 90 && bal.observations.size() == 450, "fixture sizes");
      FullBundleAdjustmentSolver bs;
      visual_navigation::scene_io::AddToSolver(&bal, &bs, 2);
      Options bo;
      bo.iteration_handle.max_num_iterations = 20;
      Summary bsum;
      EXPECT(bs.Solve(bo, &bsum), "BAL solve");
      const auto &br = bsum.GetOptimizationInfoList();
      EXPECT(!br.empty() && br.back().cost < 0.1 * br.front().cost, "BAL problem did not converge");
      if (!br.empty())
        std::printf("BAL fixture: cost %.4e -> %.4e in %zu iterations\n", br.front().cost, br.back().cost, br.size());
    }
  }
  std::printf(g_fail ? "C++ FACADE TEST FAILED (%d)\n" : "C++ FACADE TEST PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
