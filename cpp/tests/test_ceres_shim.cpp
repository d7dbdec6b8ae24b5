// CPU unit test of the Ceres-API stand-in (third_party_shim/ceres), the cost
// functors (include/core/pose_only_bundle_adjustment_solver_ceres.h) and the
// Lie-group helpers (include/utility/geometry_library.h).  No GPU.
//   * dual-number derivatives of the 6-DoF reprojection functor vs central
//     finite differences;
//   * AngleAxisRotatePoint vs the rotation matrix of so3Exp, incl. theta -> 0;
//   * the LM minimiser recovers the true pose of the noise-free pose-only scene
//     of reference test/test_compare_ceres_vs_native.cpp:20-95 (seeded), and a
//     Huber loss keeps it there with 10 % gross outliers;
//   * exp / log round trips on SO(3) and SE(3), float and double.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "ceres/ceres.h"
#include "ceres/rotation.h"
#include "core/pose_only_bundle_adjustment_solver_ceres.h"
#include "utility/geometry_library.h"

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

int main() {
  const double fx = 338, fy = 338, cx = 320, cy = 240;
  ReprojectionCostFunctor_6dof_numerical::SetCameraIntrinsicParameters(fx, fy, cx, cy);
  std::mt19937 gen(7);
  std::uniform_real_distribution<double> u(-1.0, 1.0);

  // ---- autodiff vs finite differences ----
  {
    double worst = 0;
    for (int trial = 0; trial < 50; ++trial) {
      const Eigen::Vector3d X(1.7 * u(gen), 1.3 * u(gen), 3.0 + 2.0 * u(gen));
      const Eigen::Vector2d px(320 + 100 * u(gen), 240 + 100 * u(gen));
      ceres::AutoDiffCostFunction<ReprojectionCostFunctor_6dof_numerical, 2, 6> f(
          new ReprojectionCostFunctor_6dof_numerical(X, px));
      double p[6] = {0.3 * u(gen), 0.3 * u(gen), 0.3 * u(gen), 0.2 * u(gen), 0.2 * u(gen), 0.5 * u(gen)};
      if (trial == 0) p[0] = p[1] = p[2] = 0.0;  // the theta -> 0 branch
      double r[2], J[12];
      const double *pp[1] = {p};
      double *jp[1] = {J};
      EXPECT(f.Evaluate(pp, r, jp), "Evaluate");
      for (int k = 0; k < 6; ++k) {
        const double h = 1e-6;
        double pa[6], pb[6], ra[2], rb[2];
        for (int q = 0; q < 6; ++q) pa[q] = pb[q] = p[q];
        pa[k] += h;
        pb[k] -= h;
        const double *qa[1] = {pa}, *qb[1] = {pb};
        f.Evaluate(qa, ra, nullptr);
        f.Evaluate(qb, rb, nullptr);
        for (int i = 0; i < 2; ++i) {
          const double fd = (ra[i] - rb[i]) / (2 * h);
          worst = std::fmax(worst, std::fabs(fd - J[i * 6 + k]) / std::fmax(1.0, std::fabs(fd)));
        }
      }
    }
    std::printf("autodiff vs central differences: max rel. error %.2e\n", worst);
    EXPECT(worst < 1e-6, "autodiff Jacobian");
  }
  // ---- AngleAxisRotatePoint vs so3Exp ----
  {
    double worst = 0;
    for (int trial = 0; trial < 20; ++trial) {
      const double s = trial == 0 ? 1e-12 : 1.0;
      const double w[3] = {s * u(gen), s * u(gen), s * u(gen)}, p[3] = {u(gen), u(gen), 2 + u(gen)};
      double out[3];
      ceres::AngleAxisRotatePoint(w, p, out);
      Eigen::Matrix3d R;
      geometry::so3Exp(w[0], w[1], w[2], R);
      const Eigen::Vector3d q = R * Eigen::Vector3d(p[0], p[1], p[2]);
      for (int k = 0; k < 3; ++k) worst = std::fmax(worst, std::fabs(q(k) - out[k]));
    }
    EXPECT(worst < 1e-13, "AngleAxisRotatePoint vs so3Exp: %.2e", worst);
  }
  // ---- exp / log round trips ----
  {
    double worst = 0;
    float worst_f = 0;
    for (int trial = 0; trial < 20; ++trial) {
      Eigen::Matrix<double, 6, 1> xi, back;
      for (int k = 0; k < 6; ++k) xi(k) = (trial == 0 ? 0.0 : 1.0) * u(gen);
      Eigen::Matrix<double, 4, 4> T;
      geometry::se3Exp(xi, T);
      geometry::SE3Log(T, back);
      for (int k = 0; k < 6; ++k) worst = std::fmax(worst, std::fabs(back(k) - xi(k)));
      // rotation part is orthonormal
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += T(k, a) * T(k, b);
          worst = std::fmax(worst, std::fabs(s - (a == b)));
        }
      Eigen::Matrix<float, 6, 1> xf, bf;
      for (int k = 0; k < 6; ++k) xf(k) = (float)xi(k);
      Eigen::Matrix4f Tf;
      geometry::se3Exp_f(xf, Tf);
      geometry::SE3Log_f(Tf, bf);
      for (int k = 0; k < 6; ++k) worst_f = std::fmax(worst_f, std::fabs(bf(k) - xf(k)));
      Eigen::Matrix3f Rf;
      Eigen::Matrix<float, 3, 1> wf(xf(3), xf(4), xf(5)), wb;
      geometry::so3Exp_f(wf, Rf);
      geometry::SO3Log_f(Rf, wb);
      for (int k = 0; k < 3; ++k) worst_f = std::fmax(worst_f, std::fabs(wb(k) - wf(k)));
    }
    std::printf("exp/log round trips: double %.2e, float %.2e\n", worst, worst_f);
    EXPECT(worst < 1e-12 && worst_f < 2e-5f, "exp/log round trip");
  }
  // ---- LM on the pose-only scene ----
  for (int robust = 0; robust < 2; ++robust) {
    Eigen::Isometry3d T_true = Eigen::Isometry3d::Identity();  // world -> current (reference naming)
    T_true.linear() = Eigen::AngleAxisd(-0.5, Eigen::Vector3d::UnitY()).toRotationMatrix();
    T_true.translation() = Eigen::Vector3d(0.2, 0.3, -1.9);
    const Eigen::Isometry3d Ti = T_true.inverse();
    std::vector<Eigen::Vector3d> X;
    std::vector<Eigen::Vector2d> px;
    std::uniform_real_distribution<double> dx(-1.7, 1.7), dy(-1.3, 1.3), dz(0, 5.0);
    for (int k = 0; k < 2000; ++k) {
      const Eigen::Vector3d w(dx(gen), dy(gen), dz(gen) + 1.2), l = Ti * w;
      Eigen::Vector2d p(fx * l(0) / l(2) + cx, fy * l(1) / l(2) + cy);
      if (robust && k % 10 == 0) p = p + Eigen::Vector2d(80 * u(gen), 80 * u(gen));  // gross outliers
      X.push_back(w);
      px.push_back(p);
    }
    double param[6] = {0, 0, 0, 0, 0, 0};
    ceres::Problem problem;
    for (size_t k = 0; k < X.size(); ++k)
      problem.AddResidualBlock(new ceres::AutoDiffCostFunction<ReprojectionCostFunctor_6dof_numerical, 2, 6>(
                                   new ReprojectionCostFunctor_6dof_numerical(X[k], px[k])),
                               robust ? new ceres::HuberLoss(1.0) : nullptr, param);
    ceres::Solver::Options opt;
    ceres::Solver::Summary sum;
    ceres::Solve(opt, &problem, &sum);
    std::printf("%s\n", sum.BriefReport().c_str());
    // param = (w, t) of the camera-from-world transform = T_true^-1
    Eigen::Matrix3d R;
    geometry::so3Exp(param[0], param[1], param[2], R);
    double err = 0;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) err = std::fmax(err, std::fabs(R(r, c) - Ti.linear()(r, c)));
      err = std::fmax(err, std::fabs(param[3 + r] - Ti.translation()(r)));
    }
    EXPECT(sum.termination_type == ceres::CONVERGENCE, "termination");
    EXPECT(err < (robust ? 1e-3 : 1e-8), "pose error %.3e (robust %d)", err, robust);
    EXPECT(sum.final_cost < (robust ? 1e9 : 1e-12) && sum.final_cost <= sum.initial_cost, "cost %.3e -> %.3e",
           sum.initial_cost, sum.final_cost);
  }
  std::printf(g_fail ? "CERES SHIM TEST FAILED (%d)\n" : "CERES SHIM TEST PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
