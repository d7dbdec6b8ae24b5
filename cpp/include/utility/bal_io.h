// Scene input in the "Bundle Adjustment in the Large" (BAL) text format for the
// C++ facade (the reference has no on-disk format).  Same
// mapping as bundle_adjustment_solver_amd/scene_io.py:
//   file:   P = R X + t,  p = -P / P.z,  pixel = f r(p) p,  r = 1 + k1 |p|^2 + k2 |p|^4
//   solver: one camera (f, f, 0, 0, identity extrinsics) and one pose per BAL
//           camera; T_jw = (D R, D t) with D = diag(1,-1,-1); pixel (x, -y);
//           radial distortion removed from the measurements once (exact for the
//           fixed intrinsics of this solver).
// Header-only, standard library + the facade's typedefs.
#ifndef BA_FACADE_BAL_IO_H_
#define BA_FACADE_BAL_IO_H_

#include <cmath>
#include <fstream>
#include <string>
#include <vector>

#include "core/full_bundle_adjustment_solver.h"

namespace visual_navigation {
namespace scene_io {

struct BalObservation {
  int camera = 0;  // == pose index
  int point = 0;
  analytic_solver::_BA_Pixel pixel;
};

struct BalProblem {
  std::vector<analytic_solver::_BA_Camera> cameras;  // one per BAL camera
  std::vector<analytic_solver::_BA_Pose> poses;      // camera-to-world, as AddPose expects
  std::vector<analytic_solver::_BA_Point> points;
  std::vector<BalObservation> observations;          // file order
  std::vector<double> focal, k1, k2;                 // as read
};

// false + *error on a malformed file or (undistort == false) a distorted one
inline bool LoadBal(const std::string &path, BalProblem *out, std::string *error = nullptr,
                    bool undistort = true) {
  auto fail = [&](const std::string &m) {
    if (error) *error = "BAL: " + m;
    return false;
  };
  std::ifstream in(path);
  if (!in) return fail("cannot open " + path);
  long n_cam = -1, n_pt = -1, n_obs = -1;
  if (!(in >> n_cam >> n_pt >> n_obs) || n_cam < 0 || n_pt < 0 || n_obs < 0) return fail("missing header");
  *out = BalProblem();
  out->observations.resize(n_obs);
  std::vector<double> x(n_obs), y(n_obs);
  for (long k = 0; k < n_obs; ++k) {
    long c, p;
    if (!(in >> c >> p >> x[k] >> y[k])) return fail("truncated observation list");
    if (c < 0 || c >= n_cam || p < 0 || p >= n_pt) return fail("observation refers to a camera / point out of range");
    out->observations[k].camera = static_cast<int>(c);
    out->observations[k].point = static_cast<int>(p);
  }
  out->cameras.resize(n_cam);
  out->poses.resize(n_cam);
  out->focal.resize(n_cam);
  out->k1.resize(n_cam);
  out->k2.resize(n_cam);
  bool distorted = false;
  for (long c = 0; c < n_cam; ++c) {
    double v[9];
    for (double &e : v)
      if (!(in >> e)) return fail("truncated camera block");
    // Rodrigues vector -> rotation
    const double th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double a = th < 1e-8 ? 1.0 - th * th / 6.0 : std::sin(th) / th;
    const double b = th < 1e-8 ? 0.5 - th * th / 24.0 : (1.0 - std::cos(th)) / (th * th);
    const double K[3][3] = {{0, -v[2], v[1]}, {v[2], 0, -v[0]}, {-v[1], v[0], 0}};
    double R[3][3];
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) {
        double kk = 0;
        for (int m = 0; m < 3; ++m) kk += K[r][m] * K[m][q];
        R[r][q] = (r == q ? 1.0 : 0.0) + a * K[r][q] + b * kk;
      }
    // T_jw = (D R, D t); the facade wants the inverse (camera-to-world)
    const double D[3] = {1.0, -1.0, -1.0};
    analytic_solver::_BA_Pose T_jw = analytic_solver::_BA_Pose::Identity();
    for (int r = 0; r < 3; ++r) {
      for (int q = 0; q < 3; ++q) T_jw.linear()(r, q) = D[r] * R[r][q];
      T_jw.translation()(r) = D[r] * v[3 + r];
    }
    out->poses[c] = T_jw.inverse();
    out->focal[c] = v[6];
    out->k1[c] = v[7];
    out->k2[c] = v[8];
    distorted = distorted || v[7] != 0.0 || v[8] != 0.0;
    analytic_solver::_BA_Camera &cam = out->cameras[c];
    cam.fx = cam.fy = v[6];
    cam.cx = cam.cy = 0.0;
    cam.pose_this_to_cam0 = analytic_solver::_BA_Pose::Identity();
  }
  if (distorted && !undistort) return fail("radial distortion present and undistort == false");
  out->points.resize(n_pt);
  for (long p = 0; p < n_pt; ++p) {
    double X[3];
    for (double &e : X)
      if (!(in >> e)) return fail("truncated point block");
    out->points[p] = analytic_solver::_BA_Point(X[0], X[1], X[2]);
  }
  double extra;
  if (in >> extra) return fail("trailing values");
  for (long k = 0; k < n_obs; ++k) {
    const int c = out->observations[k].camera;
    double sx = x[k], sy = y[k];
    if (out->k1[c] != 0.0 || out->k2[c] != 0.0) {  // rho (1 + k1 rho^2 + k2 rho^4) = rho_d
      const double f = out->focal[c], rd = std::sqrt(sx * sx + sy * sy) / f;
      double rho = rd;
      for (int it = 0; it < 20; ++it) {
        const double r2 = rho * rho;
        const double g = rho * (1.0 + out->k1[c] * r2 + out->k2[c] * r2 * r2) - rd;
        const double dg = 1.0 + 3.0 * out->k1[c] * r2 + 5.0 * out->k2[c] * r2 * r2;
        rho -= g / (std::fabs(dg) < 1e-12 ? 1.0 : dg);
      }
      const double s = rd > 0.0 ? rho / rd : 1.0;
      sx *= s;
      sy *= s;
    }
    out->observations[k].pixel = analytic_solver::_BA_Pixel(sx, -sy);
  }
  return true;
}

// Registers the whole problem with a solver (Add* calls; the first
// `num_fixed_poses` poses are held fixed).  `problem` must outlive the solve:
// the solver keys poses / points by their addresses and writes results back.
inline void AddToSolver(BalProblem *problem, analytic_solver::FullBundleAdjustmentSolver *solver,
                        int num_fixed_poses = 0) {
  for (size_t c = 0; c < problem->cameras.size(); ++c) solver->AddCamera(static_cast<int>(c), problem->cameras[c]);
  for (auto &T : problem->poses) solver->AddPose(&T);
  for (auto &X : problem->points) solver->AddPoint(&X);
  for (int k = 0; k < num_fixed_poses && k < static_cast<int>(problem->poses.size()); ++k)
    solver->MakePoseFixed(&problem->poses[k]);
  for (const auto &o : problem->observations)
    solver->AddObservation(o.camera, &problem->poses[o.camera], &problem->points[o.point], o.pixel);
}

}  // namespace scene_io
}  // namespace visual_navigation
#endif
