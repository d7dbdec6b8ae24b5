// GPU test: the "bundled Ceres check" of the pose-only path (BASELINE config
// C5).  Seeded form of reference test/test_compare_ceres_vs_native.cpp:73-205:
// the HIP pose-only solver behind PoseOnlyBundleAdjustmentSolver (fp32,
// analytic Jacobians, Gauss-Newton) against an INDEPENDENT fp64 estimate —
// automatic differentiation of the angle-axis reprojection functor
// (core/pose_only_bundle_adjustment_solver_ceres.h) minimised by the
// Levenberg-Marquardt of the Ceres-API stand-in — on 10 000 points (config C5)
// and 300 000 points (the reference demo's size), noise-free and with
// sigma = 0.5 px pixel noise.  Poses must agree to 1e-3.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "ceres/ceres.h"
#include "ceres/rotation.h"
#include "core/pose_only_bundle_adjustment_solver.h"
#include "core/pose_only_bundle_adjustment_solver_ceres.h"
#include "utility/geometry_library.h"
#include "utility/timer.h"

using Pose = Eigen::Isometry3f;
using Position = Eigen::Vector3f;
using Pixel = Eigen::Vector2f;
using namespace visual_navigation::analytic_solver;

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

static void PoseDiff(const Pose &a, const Pose &b, float *ang, float *dt) {
  const Eigen::Matrix3f D = a.linear() * b.linear().transpose();
  const float wx = 0.5f * (D(2, 1) - D(1, 2)), wy = 0.5f * (D(0, 2) - D(2, 0)), wz = 0.5f * (D(1, 0) - D(0, 1));
  *ang = std::asin(std::fmin(1.0f, std::sqrt(wx * wx + wy * wy + wz * wz)));
  *dt = (a.translation() - b.translation()).norm();
}

int main() {
  const float fx = 338.0f, fy = 338.0f, cx = 320.0f, cy = 240.0f;
  Pose pose_world_to_current = Pose::Identity();
  pose_world_to_current.linear() = Eigen::AngleAxisf(-0.5f, Position::UnitY()).toRotationMatrix();
  pose_world_to_current.translation() = Position(0.2f, 0.3f, -1.9f);
  PoseOnlyBundleAdjustmentSolver native;
  for (const int num_points : {10000, 300000})
    for (const float sigma : {0.0f, 0.5f}) {
      std::mt19937 gen(20240605u + num_points);
      std::uniform_real_distribution<float> dist_x(-1.7f, 1.7f), dist_y(-1.3f, 1.3f), dist_z(0.0f, 5.0f);
      std::normal_distribution<float> dist_pixel(0.0f, 1.0f);
      std::vector<Position> world;
      std::vector<Pixel> pixels;
      const Pose Ti = pose_world_to_current.inverse();
      for (int k = 0; k < num_points; ++k) {
        const Position w(dist_x(gen), dist_y(gen), dist_z(gen) + 1.2f), l = Ti * w;
        const float iz = 1.0f / l.z();
        world.push_back(w);
        pixels.push_back(Pixel(fx * l.x() * iz + cx + sigma * dist_pixel(gen), fy * l.y() * iz + cy + sigma * dist_pixel(gen)));
      }
      // 1) native (HIP) solver, options of the reference demo (:131-141)
      Pose pose_native = Pose::Identity();
      Options options;
      options.iteration_handle.max_num_iterations = 100;
      options.convergence_handle.threshold_cost_change = 1e-6f;
      options.convergence_handle.threshold_step_size = 1e-6f;
      options.outlier_handle.threshold_huber_loss = 1.0f;
      options.outlier_handle.threshold_outlier_rejection = 2.5f;
      options.solver_type = SolverType::GAUSS_NEWTON;
      Summary summary_native;
      std::vector<bool> mask;
      timer::tic();
      EXPECT(native.Solve_Monocular_6Dof(world, pixels, fx, fy, cx, cy, pose_native, mask, options, &summary_native),
             "native solve");
      const double ms_native = timer::toc(0);
      // 2) fp64 autodiff + LM (:178-205)
      timer::tic();
      double param[6] = {0, 0, 0, 0, 0, 0};
      ReprojectionCostFunctor_6dof_numerical::SetCameraIntrinsicParameters(fx, fy, cx, cy);
      ceres::Problem problem;
      for (int k = 0; k < num_points; ++k)
        problem.AddResidualBlock(new ceres::AutoDiffCostFunction<ReprojectionCostFunctor_6dof_numerical, 2, 6>(
                                     new ReprojectionCostFunctor_6dof_numerical(world[k].cast<double>(), pixels[k].cast<double>())),
                                 nullptr, param);
      ceres::Solver::Options copt;
      ceres::Solver::Summary csum;
      ceres::Solve(copt, &problem, &csum);
      const double ms_autodiff = timer::toc(0);
      Eigen::Matrix<float, 3, 1> w_c2w((float)param[0], (float)param[1], (float)param[2]);
      Eigen::Matrix3f R_c2w;
      geometry::so3Exp_f(w_c2w, R_c2w);
      const Position t_c2w((float)param[3], (float)param[4], (float)param[5]);
      Pose pose_autodiff = Pose::Identity();
      pose_autodiff.linear() = R_c2w.transpose();
      pose_autodiff.translation() = -(R_c2w.transpose() * t_c2w);
      float ang, dt, ang_t, dt_t;
      PoseDiff(pose_native, pose_autodiff, &ang, &dt);
      PoseDiff(pose_native, pose_world_to_current, &ang_t, &dt_t);
      std::printf("%6d points, sigma %.1f px: native vs autodiff-LM: angle %.2e rad, |dt| %.2e m; native vs truth: %.2e rad, "
                  "%.2e m; %s; native %.2f ms (%zu rows), autodiff-LM %.0f ms\n",
                  num_points, sigma, ang, dt, ang_t, dt_t, csum.BriefReport().c_str(), ms_native,
                  summary_native.GetOptimizationInfoList().size(), ms_autodiff);
      EXPECT(csum.termination_type == ceres::CONVERGENCE, "autodiff LM did not converge");
      EXPECT(ang <= 1e-3f && dt <= 1e-3f, "HIP pose-only differs from the autodiff-LM estimate");
      EXPECT(sigma > 0.0f || (ang_t <= 1e-3f && dt_t <= 1e-3f), "noise-free: truth not recovered");
    }
  std::printf(g_fail ? "COMPARE TEST FAILED (%d)\n" : "COMPARE TEST PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
