// ba_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).
//
// Dependency-free restatement of the reference's full-BA LM loop and of the
// pose-only monocular 6-DoF Gauss-Newton loop.  Every function cites the
// reference lines it follows (paths relative to the reference root).
//
// PARITY UNPINNED: the reference ships no golden vectors and cannot be built
// in this image (Eigen / Ceres / OpenCV absent).  The
// third-party arithmetic restated here is Eigen3 (version unpinned by the
// reference's CMakeLists.txt:9): fixed-size products (summed in k order) and
// LDLT (`ldlt_inplace<Lower>::unblocked` + `LDLT::_solve_impl`, Eigen 3.4
// behaviour: diagonal pivoting, D pseudo-inverted with tolerance DBL_MIN).
//
// Deliberate, documented deviations from the reference:
//  * optimisation indices follow INPUT order of the non-fixed entries (the
//    reference uses unordered_map iteration order, i.e. pointer-hash order);
//  * per-landmark pose sets are iterated in ascending j (the reference
//    iterates unordered_set<int> order).
// Both only permute floating-point summation order.
//
// Build: see oracle/Makefile (g++ -O2 -std=c++17, no -ffast-math).

#include "ba_oracle.h"

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

using std::size_t;

struct Cam {
  double fx, fy, cx, cy;
  double R[9];
  double t[3];
};

struct Pose {
  double R[9];
  double t[3];
};

inline double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(
             high_resolution_clock::now().time_since_epoch())
      .count();
}

// ---------------------------------------------------------------------------
// Eigen-style pivoted LDLT (lower), restating Eigen 3.4
// internal::ldlt_inplace<Lower>::unblocked and LDLT::_solve_impl.
// `a` is n x n row-major, only the lower triangle is read / written.
// ---------------------------------------------------------------------------
struct Ldlt {
  int n = 0;
  std::vector<double> m;   // row-major n x n (lower = L strictly, diag = D)
  std::vector<int> tr;     // transpositions
  std::vector<double> tmp;

  inline double &at(int r, int c) { return m[(size_t)r * n + c]; }

  void compute(int n_, const double *a_rowmajor) {
    n = n_;
    m.assign(a_rowmajor, a_rowmajor + (size_t)n * n);
    tr.resize(n);
    tmp.resize(n);
    if (n <= 1) {
      if (n == 1) tr[0] = 0;
      return;
    }
    for (int k = 0; k < n; ++k) {
      // biggest |diag| in the remaining corner
      int big = k;
      double bigv = std::fabs(at(k, k));
      for (int i = k + 1; i < n; ++i) {
        double v = std::fabs(at(i, i));
        if (v > bigv) {
          bigv = v;
          big = i;
        }
      }
      tr[k] = big;
      if (k != big) {
        const int s = n - big - 1;
        for (int c = 0; c < k; ++c) std::swap(at(k, c), at(big, c));
        for (int r = 0; r < s; ++r)
          std::swap(at(big + 1 + r, k), at(big + 1 + r, big));
        std::swap(at(k, k), at(big, big));
        for (int i = k + 1; i < big; ++i) {
          double t = at(i, k);
          at(i, k) = at(big, i);
          at(big, i) = t;
        }
      }
      const int rs = n - k - 1;
      if (k > 0) {
        // temp.head(k) = D.head(k) .* A10^T ; A10 = row k, cols [0,k)
        double acc = 0.0;
        for (int c = 0; c < k; ++c) {
          tmp[c] = at(c, c) * at(k, c);
          acc += at(k, c) * tmp[c];
        }
        at(k, k) -= acc;
        // A21 -= A20 * temp
        // (4 partial sums, as Eigen's packetised gemv reduces: keeps the
        // CPU baseline from being latency-bound on one FMA chain)
        for (int r = 0; r < rs; ++r) {
          double *row = &m[(size_t)(k + 1 + r) * n];
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
          int c = 0;
          for (; c + 4 <= k; c += 4) {
            s0 += row[c] * tmp[c];
            s1 += row[c + 1] * tmp[c + 1];
            s2 += row[c + 2] * tmp[c + 2];
            s3 += row[c + 3] * tmp[c + 3];
          }
          for (; c < k; ++c) s0 += row[c] * tmp[c];
          row[k] -= (s0 + s1) + (s2 + s3);
        }
      }
      const double akk = at(k, k);
      const bool pivot_valid = (std::fabs(akk) > 0.0);
      if (k == 0 && !pivot_valid) {
        // whole diagonal is zero -> Eigen stops with identity transpositions
        for (int j = 0; j < n; ++j) tr[j] = j;
        return;
      }
      if (rs > 0 && pivot_valid)
        for (int r = 0; r < rs; ++r) at(k + 1 + r, k) /= akk;
    }
  }

  // solve for one right-hand side in place
  void solve(double *b) const {
    const int nn = n;
    // dst = P b
    for (int i = 0; i < nn; ++i)
      if (tr[i] != i) std::swap(b[i], b[tr[i]]);
    // L solve (unit lower)
    for (int i = 0; i < nn; ++i) {
      const double *row = &m[(size_t)i * nn];
      double s = b[i];
      for (int c = 0; c < i; ++c) s -= row[c] * b[c];
      b[i] = s;
    }
    // D pseudo-inverse
    const double tol = DBL_MIN;
    for (int i = 0; i < nn; ++i) {
      const double d = m[(size_t)i * nn + i];
      if (std::fabs(d) > tol)
        b[i] /= d;
      else
        b[i] = 0.0;
    }
    // L^T solve
    for (int i = nn - 1; i >= 0; --i) {
      double s = b[i];
      for (int r = i + 1; r < nn; ++r) s -= m[(size_t)r * nn + i] * b[r];
      b[i] = s;
    }
    // dst = P^T dst
    for (int i = nn - 1; i >= 0; --i)
      if (tr[i] != i) std::swap(b[i], b[tr[i]]);
  }
};

// 3x3 specialisation of the same algorithm: Cinv = C.ldlt().solve(I)
// (reference core/full_bundle_adjustment_solver.cpp:854).
void ldlt3_inverse(const double C[9], double Cinv[9]) {
  Ldlt f;
  f.compute(3, C);
  for (int c = 0; c < 3; ++c) {
    double e[3] = {0, 0, 0};
    e[c] = 1.0;
    f.solve(e);
    for (int r = 0; r < 3; ++r) Cinv[r * 3 + c] = e[r];
  }
}

// ---------------------------------------------------------------------------
// FAST-SOLVE MODE (test infrastructure only, NOT a restatement of the
// reference): envelope ("skyline") LDL^T WITHOUT pivoting of the same reduced
// system S x = rhs that reference :905 hands to Eigen's pivoted LDLT.  S is
// block sparse (two poses couple only through common landmarks), so row r of
// the factor has no entries left of first[r] = the first structurally
// non-zero column of row r of S; the work drops from n^3/3 to sum_r w_r^2
// (C4: 71 GFLOP -> ~10 MFLOP) and the BASELINE-size trajectories can be
// followed for many iterations.  S is symmetric positive definite whenever the
// pivoted factorisation meets no zero pivot, and then both give the same x up
// to roundoff (tests/test_oracle_pins.py cross-checks them to 1e-9 on C1 and
// C2).  A zero pivot (pose without observations: zero row and column) yields
// x = 0 for that unknown, as the pseudo-inverted D of Eigen's solve does.
// ---------------------------------------------------------------------------
struct Skyline {
  int n = 0;
  std::vector<int> first;        // first stored column of row r (<= r)
  std::vector<size_t> rowp;      // offset of S(r, first[r]) in `v`
  std::vector<double> v;         // rows of the envelope, diagonal included
  std::vector<double> t;

  // a: n x n row-major, lower triangle read
  void compute(int n_, const double *a) {
    n = n_;
    first.resize(n);
    rowp.resize(n + 1);
    size_t tot = 0;
    for (int r = 0; r < n; ++r) {
      const double *row = a + (size_t)r * n;
      int f = r;
      for (int c = 0; c < r; ++c)
        if (row[c] != 0.0) {
          f = c;
          break;
        }
      first[r] = f;
      rowp[r] = tot;
      tot += (size_t)(r - f + 1);
    }
    rowp[n] = tot;
    v.resize(tot);
    t.resize(n);
    for (int r = 0; r < n; ++r)
      std::memcpy(&v[rowp[r]], a + (size_t)r * n + first[r],
                  sizeof(double) * (size_t)(r - first[r] + 1));
    // row by row: t_c = L(r,c) D_c, then L(r,c) = t_c / D_c, D_r
    for (int r = 0; r < n; ++r) {
      double *Lr = &v[rowp[r]] - first[r];  // Lr[c], first[r] <= c <= r
      for (int c = first[r]; c < r; ++c) {
        const double *Lc = &v[rowp[c]] - first[c];
        const int k0 = std::max(first[r], first[c]);
        double s = Lr[c];
        for (int k = k0; k < c; ++k) s -= t[k] * Lc[k];
        t[c] = s;
      }
      double d = Lr[r];
      for (int c = first[r]; c < r; ++c) {
        const double dc = v[rowp[c] + (size_t)(c - first[c])];
        const double l = (std::fabs(dc) > 0.0) ? t[c] / dc : t[c];
        Lr[c] = l;
        d -= t[c] * l;
      }
      Lr[r] = d;
    }
  }

  void solve(double *b) const {
    for (int r = 0; r < n; ++r) {  // L z = b
      const double *Lr = &v[rowp[r]] - first[r];
      double s = b[r];
      for (int c = first[r]; c < r; ++c) s -= Lr[c] * b[c];
      b[r] = s;
    }
    for (int r = 0; r < n; ++r) {  // D pseudo-inverse
      const double d = v[rowp[r] + (size_t)(r - first[r])];
      b[r] = (std::fabs(d) > DBL_MIN) ? b[r] / d : 0.0;
    }
    for (int r = n - 1; r >= 0; --r) {  // L^T x = w
      const double *Lr = &v[rowp[r]] - first[r];
      const double xr = b[r];
      for (int c = first[r]; c < r; ++c) b[c] -= Lr[c] * xr;
    }
  }
};

// se3 exponential, reference core/full_bundle_adjustment_solver.cpp:1046-1082
template <typename T>
void se3_exp(const T xi[6], T R[9], T t[3]) {
  const T v[3] = {xi[0], xi[1], xi[2]};
  const T w[3] = {xi[3], xi[4], xi[5]};
  const T theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const T wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  T wx2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      T s = 0;
      for (int k = 0; k < 3; ++k) s += wx[r * 3 + k] * wx[k * 3 + c];
      wx2[r * 3 + c] = s;
    }
  T V[9];
  T ca, cb, va, vb;
  if (theta < (T)1e-7) {
    ca = (T)1;
    cb = (T)0.5;
    va = (T)0.5;
    vb = (T)0.33333333333333333333333333;
  } else {
    ca = std::sin(theta) / theta;
    cb = ((T)1 - std::cos(theta)) / (theta * theta);
    va = cb;
    vb = (theta - std::sin(theta)) / (theta * theta * theta);
  }
  for (int i = 0; i < 9; ++i) {
    const T id = (i % 4 == 0) ? (T)1 : (T)0;
    R[i] = id + ca * wx[i] + cb * wx2[i];
    V[i] = id + va * wx[i] + vb * wx2[i];
  }
  for (int r = 0; r < 3; ++r)
    t[r] = V[r * 3 + 0] * v[0] + V[r * 3 + 1] * v[1] + V[r * 3 + 2] * v[2];
}

template <typename T>
void rigid_compose(const T Ra[9], const T ta[3], const T Rb[9], const T tb[3],
                   T Rout[9], T tout[3]) {
  // (Ra,ta) * (Rb,tb)
  T Rn[9], tn[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      T s = 0;
      for (int k = 0; k < 3; ++k) s += Ra[r * 3 + k] * Rb[k * 3 + c];
      Rn[r * 3 + c] = s;
    }
    tn[r] = Ra[r * 3 + 0] * tb[0] + Ra[r * 3 + 1] * tb[1] +
            Ra[r * 3 + 2] * tb[2] + ta[r];
  }
  std::memcpy(Rout, Rn, sizeof(Rn));
  std::memcpy(tout, tn, sizeof(tn));
}

}  // namespace

// ===========================================================================
struct ba_oracle {
  // problem
  std::vector<Cam> cams;
  std::vector<Pose> poses;          // T_jw, all poses
  std::vector<uint8_t> pose_fixed;
  std::vector<double> X;            // all points, 3 each
  std::vector<uint8_t> pt_fixed;
  std::vector<int32_t> ocam, opose, opt_;
  std::vector<double> ouv;
  int64_t n_obs = 0;

  // index maps (reference FinalizeParameters :182-206, input order)
  int N = 0, M = 0;
  std::vector<int> j_opt, i_opt;    // per pose / point, -1 if fixed
  std::vector<int> pose_of_j, pt_of_i;

  // connectivity (reference :668-700): pairs sorted by (i, j)
  std::vector<int64_t> lm_ptr;      // M+1
  std::vector<int32_t> pair_j;      // P
  std::vector<int64_t> obs_pair;    // per observation, -1 if not both opt

  // storage
  std::vector<double> A, a;         // N*36, N*6
  std::vector<double> C, b;         // M*9,  M*3
  std::vector<double> W;            // P*18  (B_ji, 6x3 row-major)
  std::vector<double> Cinv, Cinvb;  // M*9, M*3
  std::vector<double> V;            // P*18  (BCinv_ji)
  std::vector<double> BCinvb;       // N*6
  std::vector<double> S, rhs, x;    // (6N)^2, 6N, 6N
  std::vector<double> y;            // M*3
  std::vector<Pose> poses_bak;
  std::vector<double> X_bak;

  bool dense_faithful = false;
  bool fast_solve = false;          // Skyline instead of the pivoted LDLT
  std::vector<double> dense_grid;   // 4 * N*M*18 doubles when dense_faithful

  double stage_ms[4] = {0, 0, 0, 0};

  void build_structure();
};

void ba_oracle::build_structure() {
  const int n_pose = (int)poses.size();
  const int n_pt = (int)pt_fixed.size();
  j_opt.assign(n_pose, -1);
  i_opt.assign(n_pt, -1);
  N = 0;
  for (int p = 0; p < n_pose; ++p)
    if (!pose_fixed[p]) {
      j_opt[p] = N++;
      pose_of_j.push_back(p);
    }
  M = 0;
  for (int q = 0; q < n_pt; ++q)
    if (!pt_fixed[q]) {
      i_opt[q] = M++;
      pt_of_i.push_back(q);
    }
  // distinct (i, j) pairs with both optimisable
  std::vector<std::pair<int32_t, int32_t>> pr;
  pr.reserve(n_obs);
  for (int64_t k = 0; k < n_obs; ++k) {
    const int i = i_opt[opt_[k]], j = j_opt[opose[k]];
    if (i >= 0 && j >= 0) pr.emplace_back(i, j);
  }
  std::sort(pr.begin(), pr.end());
  pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
  const int64_t P = (int64_t)pr.size();
  lm_ptr.assign(M + 1, 0);
  pair_j.resize(P);
  for (int64_t p = 0; p < P; ++p) {
    lm_ptr[pr[p].first + 1]++;
    pair_j[p] = pr[p].second;
  }
  for (int i = 0; i < M; ++i) lm_ptr[i + 1] += lm_ptr[i];
  obs_pair.assign(n_obs, -1);
  for (int64_t k = 0; k < n_obs; ++k) {
    const int i = i_opt[opt_[k]], j = j_opt[opose[k]];
    if (i >= 0 && j >= 0) {
      const int32_t *b0 = &pair_j[lm_ptr[i]], *e0 = &pair_j[lm_ptr[i + 1]];
      obs_pair[k] = (std::lower_bound(b0, e0, j) - &pair_j[0]);
    }
  }
  A.assign((size_t)N * 36, 0);
  a.assign((size_t)N * 6, 0);
  C.assign((size_t)M * 9, 0);
  b.assign((size_t)M * 3, 0);
  W.assign((size_t)P * 18, 0);
  Cinv.assign((size_t)M * 9, 0);
  Cinvb.assign((size_t)M * 3, 0);
  V.assign((size_t)P * 18, 0);
  BCinvb.assign((size_t)N * 6, 0);
  S.assign((size_t)36 * N * N, 0);
  rhs.assign((size_t)6 * N, 0);
  x.assign((size_t)6 * N, 0);
  y.assign((size_t)M * 3, 0);
}

// Residual of one observation at the current parameters.
// reference core/full_bundle_adjustment_solver.cpp:402-425 and :733-760.
// The reference writes the projection in two ways that differ in the last bit:
// the linearisation forms `fx * xinvz + cx` with xinvz = xj * invz (:753, :759),
// EvaluateCurrentCost forms `fx * xj * invz + cx`, i.e. (fx * xj) * invz (:424).
// COST selects the second form (ba_oracle_cost only).
template <bool COST = false>
static inline void project(const ba_oracle *o, int64_t k, double Xij[3],
                           double Xc[3], double r[2]) {
  const Cam &cam = o->cams[o->ocam[k]];
  const Pose &T = o->poses[o->opose[k]];
  const double *Xi = &o->X[(size_t)o->opt_[k] * 3];
  for (int r0 = 0; r0 < 3; ++r0)
    Xij[r0] = (T.R[r0 * 3 + 0] * Xi[0] + T.R[r0 * 3 + 1] * Xi[1] +
               T.R[r0 * 3 + 2] * Xi[2]) +
              T.t[r0];
  for (int r0 = 0; r0 < 3; ++r0)
    Xc[r0] = (cam.R[r0 * 3 + 0] * Xij[0] + cam.R[r0 * 3 + 1] * Xij[1] +
              cam.R[r0 * 3 + 2] * Xij[2]) +
             cam.t[r0];
  const double invz = 1.0 / Xc[2];
  if (COST) {
    r[0] = (cam.fx * Xc[0]) * invz + cam.cx - o->ouv[2 * k + 0];
    r[1] = (cam.fy * Xc[1]) * invz + cam.cy - o->ouv[2 * k + 1];
  } else {
    r[0] = cam.fx * (Xc[0] * invz) + cam.cx - o->ouv[2 * k + 0];
    r[1] = cam.fy * (Xc[1] * invz) + cam.cy - o->ouv[2 * k + 1];
  }
}

extern "C" {

ba_oracle *ba_oracle_create(int n_cam, const double *cam_intr4,
                            const double *cam_T12, int n_pose,
                            const double *pose_T12, const uint8_t *pose_fixed,
                            int n_pt, const double *pt_X3,
                            const uint8_t *pt_fixed, int64_t n_obs,
                            const int32_t *obs_cam, const int32_t *obs_pose,
                            const int32_t *obs_pt, const double *obs_uv2) {
  ba_oracle *o = new ba_oracle();
  o->cams.resize(n_cam);
  for (int c = 0; c < n_cam; ++c) {
    Cam &cm = o->cams[c];
    cm.fx = cam_intr4[4 * c + 0];
    cm.fy = cam_intr4[4 * c + 1];
    cm.cx = cam_intr4[4 * c + 2];
    cm.cy = cam_intr4[4 * c + 3];
    std::memcpy(cm.R, cam_T12 + 12 * c, 9 * sizeof(double));
    std::memcpy(cm.t, cam_T12 + 12 * c + 9, 3 * sizeof(double));
  }
  o->poses.resize(n_pose);
  for (int p = 0; p < n_pose; ++p) {
    std::memcpy(o->poses[p].R, pose_T12 + 12 * p, 9 * sizeof(double));
    std::memcpy(o->poses[p].t, pose_T12 + 12 * p + 9, 3 * sizeof(double));
  }
  o->pose_fixed.assign(pose_fixed, pose_fixed + n_pose);
  o->X.assign(pt_X3, pt_X3 + (size_t)n_pt * 3);
  o->pt_fixed.assign(pt_fixed, pt_fixed + n_pt);
  o->n_obs = n_obs;
  o->ocam.assign(obs_cam, obs_cam + n_obs);
  o->opose.assign(obs_pose, obs_pose + n_obs);
  o->opt_.assign(obs_pt, obs_pt + n_obs);
  o->ouv.assign(obs_uv2, obs_uv2 + 2 * n_obs);
  o->build_structure();
  return o;
}

void ba_oracle_destroy(ba_oracle *o) { delete o; }

void ba_oracle_set_dense_faithful(ba_oracle *o, int on) {
  o->dense_faithful = on != 0;
  if (o->dense_faithful)
    o->dense_grid.assign((size_t)4 * o->N * o->M * 18, 0.0);
  else
    std::vector<double>().swap(o->dense_grid);
}

void ba_oracle_set_fast_solve(ba_oracle *o, int on) { o->fast_solve = on != 0; }

int ba_oracle_num_opt_poses(const ba_oracle *o) { return o->N; }
int ba_oracle_num_opt_points(const ba_oracle *o) { return o->M; }
int64_t ba_oracle_num_pairs(const ba_oracle *o) {
  return (int64_t)o->pair_j.size();
}

// reference :381-433 — sum of UNSQUARED residual norms over all observations
double ba_oracle_cost(ba_oracle *o) {
  double err = 0.0;
  for (int64_t k = 0; k < o->n_obs; ++k) {
    double Xij[3], Xc[3], r[2];
    project<true>(o, k, Xij, Xc, r);
    err += std::sqrt(r[0] * r[0] + r[1] * r[1]);
  }
  return err;
}

// reference :343-379 (reset) + :716-831 (per-observation linearisation)
void ba_oracle_linearize(ba_oracle *o, double huber) {
  std::fill(o->A.begin(), o->A.end(), 0.0);
  std::fill(o->a.begin(), o->a.end(), 0.0);
  std::fill(o->C.begin(), o->C.end(), 0.0);
  std::fill(o->b.begin(), o->b.end(), 0.0);
  std::fill(o->W.begin(), o->W.end(), 0.0);
  std::fill(o->V.begin(), o->V.end(), 0.0);
  std::fill(o->BCinvb.begin(), o->BCinvb.end(), 0.0);
  std::fill(o->S.begin(), o->S.end(), 0.0);
  std::fill(o->x.begin(), o->x.end(), 0.0);
  std::fill(o->y.begin(), o->y.end(), 0.0);
  if (o->dense_faithful)  // reference :352-357: four dense grids re-zeroed
    std::fill(o->dense_grid.begin(), o->dense_grid.end(), 0.0);

  for (int64_t k = 0; k < o->n_obs; ++k) {
    const Cam &cam = o->cams[o->ocam[k]];
    const Pose &T = o->poses[o->opose[k]];
    const int j = o->j_opt[o->opose[k]];
    const int i = o->i_opt[o->opt_[k]];
    double Xij[3], Xc[3], r[2];
    project(o, k, Xij, Xc, r);
    // :750-754
    const double invz = 1.0 / Xc[2];
    const double fxinvz = cam.fx * invz, fyinvz = cam.fy * invz;
    const double xinvz = Xc[0] * invz, yinvz = Xc[1] * invz;
    const double fx_xinvz2 = fxinvz * xinvz, fy_yinvz2 = fyinvz * yinvz;
    // :763-768  (floating abs)
    const double absrxry = std::fabs(r[0]) + std::fabs(r[1]);
    const double weight = (absrxry > huber) ? (huber / absrxry) : 1.0;
    const double wr[2] = {weight * r[0], weight * r[1]};
    // :770-787  G = dpi/dXc * R_cj
    double G[6];
    G[0] = fxinvz * cam.R[0] + (-fx_xinvz2) * cam.R[6];
    G[1] = fxinvz * cam.R[1] + (-fx_xinvz2) * cam.R[7];
    G[2] = fxinvz * cam.R[2] + (-fx_xinvz2) * cam.R[8];
    G[3] = fyinvz * cam.R[3] + (-fy_yinvz2) * cam.R[6];
    G[4] = fyinvz * cam.R[4] + (-fy_yinvz2) * cam.R[7];
    G[5] = fyinvz * cam.R[5] + (-fy_yinvz2) * cam.R[8];
    double Q[12];  // 2x6 row-major
    if (j >= 0) {
      // :797-800  Q = [G, G * (-[Xij]x)]
      const double Sk[9] = {0.0,     Xij[2],  -Xij[1], -Xij[2], 0.0,
                            Xij[0],  Xij[1],  -Xij[0], 0.0};
      for (int rr = 0; rr < 2; ++rr) {
        for (int c = 0; c < 3; ++c) Q[rr * 6 + c] = G[rr * 3 + c];
        for (int c = 0; c < 3; ++c)
          Q[rr * 6 + 3 + c] = G[rr * 3 + 0] * Sk[0 * 3 + c] +
                              G[rr * 3 + 1] * Sk[1 * 3 + c] +
                              G[rr * 3 + 2] * Sk[2 * 3 + c];
      }
      // :519-556 upper triangle of w Q^T Q; :558-598 add
      double *Aj = &o->A[(size_t)j * 36];
      for (int rr = 0; rr < 6; ++rr)
        for (int c = rr; c < 6; ++c)
          Aj[rr * 6 + c] += (weight * Q[rr]) * Q[c] +
                            (weight * Q[6 + rr]) * Q[6 + c];
      // :809
      double *aj = &o->a[(size_t)j * 6];
      for (int c = 0; c < 6; ++c) aj[c] -= Q[c] * wr[0] + Q[6 + c] * wr[1];
    }
    if (i >= 0) {
      // :814 R = G * R_jw
      double Rm[6];
      for (int rr = 0; rr < 2; ++rr)
        for (int c = 0; c < 3; ++c)
          Rm[rr * 3 + c] = G[rr * 3 + 0] * T.R[0 * 3 + c] +
                           G[rr * 3 + 1] * T.R[1 * 3 + c] +
                           G[rr * 3 + 2] * T.R[2 * 3 + c];
      // :503-517, :558-568
      double *Ci = &o->C[(size_t)i * 9];
      for (int rr = 0; rr < 3; ++rr)
        for (int c = rr; c < 3; ++c)
          Ci[rr * 3 + c] +=
              weight * (Rm[rr] * Rm[c] + Rm[3 + rr] * Rm[3 + c]);
      // :823
      double *bi = &o->b[(size_t)i * 3];
      for (int c = 0; c < 3; ++c) bi[c] -= Rm[c] * wr[0] + Rm[3 + c] * wr[1];
      if (j >= 0) {
        // :826  B_ji = weight * (Q^T R)  — ASSIGNMENT (last writer wins, Q1)
        double *Wp = &o->W[(size_t)o->obs_pair[k] * 18];
        for (int rr = 0; rr < 6; ++rr)
          for (int c = 0; c < 3; ++c)
            Wp[rr * 3 + c] =
                weight * (Q[rr] * Rm[c] + Q[6 + rr] * Rm[3 + c]);
      }
    }
  }
}

// reference :833-856
void ba_oracle_damp_invert(ba_oracle *o, double lambda) {
  const double lp1 = 1.0 + lambda;
  for (int j = 0; j < o->N; ++j) {
    double *Aj = &o->A[(size_t)j * 36];
    for (int r = 0; r < 6; ++r)
      for (int c = r + 1; c < 6; ++c) Aj[c * 6 + r] = Aj[r * 6 + c];
    for (int r = 0; r < 6; ++r) Aj[r * 6 + r] *= lp1;
  }
  for (int i = 0; i < o->M; ++i) {
    double *Ci = &o->C[(size_t)i * 9];
    Ci[3] = Ci[1];
    Ci[6] = Ci[2];
    Ci[7] = Ci[5];
    Ci[0] *= lp1;
    Ci[4] *= lp1;
    Ci[8] *= lp1;
    double *Ii = &o->Cinv[(size_t)i * 9];
    ldlt3_inverse(Ci, Ii);
    const double *bi = &o->b[(size_t)i * 3];
    double *cb = &o->Cinvb[(size_t)i * 3];
    for (int r = 0; r < 3; ++r)
      cb[r] = Ii[r * 3 + 0] * bi[0] + Ii[r * 3 + 1] * bi[1] +
              Ii[r * 3 + 2] * bi[2];
  }
}

// reference :858-888
void ba_oracle_schur(ba_oracle *o) {
  const int N = o->N;
  const int n6 = 6 * N;
  std::vector<double> BCB((size_t)36 * N * N, 0.0);  // BCinvBt_, upper blocks
  std::fill(o->BCinvb.begin(), o->BCinvb.end(), 0.0);
  for (int i = 0; i < o->M; ++i) {
    const double *Ii = &o->Cinv[(size_t)i * 9];
    const double *bi = &o->b[(size_t)i * 3];
    const int64_t p0 = o->lm_ptr[i], p1 = o->lm_ptr[i + 1];
    for (int64_t p = p0; p < p1; ++p) {
      const int j = o->pair_j[p];
      const double *Wj = &o->W[(size_t)p * 18];
      double *Vj = &o->V[(size_t)p * 18];
      // :862 BCinv_ji = B_ji * Cinv_i
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 3; ++c)
          Vj[r * 3 + c] = Wj[r * 3 + 0] * Ii[0 * 3 + c] +
                          Wj[r * 3 + 1] * Ii[1 * 3 + c] +
                          Wj[r * 3 + 2] * Ii[2 * 3 + c];
      // :864 BCinv_b_j += BCinv_ji * b_i
      double *bj = &o->BCinvb[(size_t)j * 6];
      for (int r = 0; r < 6; ++r)
        bj[r] += Vj[r * 3 + 0] * bi[0] + Vj[r * 3 + 1] * bi[1] +
                 Vj[r * 3 + 2] * bi[2];
      // :866-870 BCinvBt_jk += BCinv_ji * Bt_ik, k >= j
      for (int64_t q = p0; q < p1; ++q) {
        const int k = o->pair_j[q];
        if (k < j) continue;
        const double *Wk = &o->W[(size_t)q * 18];
        double *blk = &BCB[((size_t)j * N + k) * 36];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c)
            blk[r * 6 + c] += Vj[r * 3 + 0] * Wk[c * 3 + 0] +
                              Vj[r * 3 + 1] * Wk[c * 3 + 1] +
                              Vj[r * 3 + 2] * Wk[c * 3 + 2];
      }
    }
  }
  // :874-888 mirror, S = A - BCinvBt, rhs = a - BCinv_b ; :892-902 dense copy
  for (int j = 0; j < N; ++j)
    for (int k = 0; k < N; ++k) {
      const double *blk = (k >= j) ? &BCB[((size_t)j * N + k) * 36]
                                   : &BCB[((size_t)k * N + j) * 36];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
          const double v = (k >= j) ? blk[r * 6 + c] : blk[c * 6 + r];
          const double av = (j == k) ? o->A[(size_t)j * 36 + r * 6 + c] : 0.0;
          o->S[(size_t)(6 * j + r) * n6 + 6 * k + c] =
              (j == k) ? (av - v) : (-v);
        }
    }
  for (int j = 0; j < N; ++j)
    for (int r = 0; r < 6; ++r)
      o->rhs[6 * j + r] = o->a[6 * j + r] - o->BCinvb[6 * j + r];
}

// reference :905  x = S.ldlt().solve(rhs)
void ba_oracle_solve_reduced(ba_oracle *o) {
  const int n6 = 6 * o->N;
  o->x = o->rhs;
  if (o->fast_solve) {  // test-only shortcut, see Skyline above
    Skyline f;
    f.compute(n6, o->S.data());
    f.solve(o->x.data());
    return;
  }
  Ldlt f;
  f.compute(n6, o->S.data());
  f.solve(o->x.data());
}

// reference :910-917  y_i = Cinv_b_i - sum_j CinvBt_ij x_j
void ba_oracle_backsub(ba_oracle *o) {
  for (int i = 0; i < o->M; ++i) {
    double acc[3] = {0, 0, 0};
    for (int64_t p = o->lm_ptr[i]; p < o->lm_ptr[i + 1]; ++p) {
      const int j = o->pair_j[p];
      const double *Vj = &o->V[(size_t)p * 18];
      const double *xj = &o->x[(size_t)6 * j];
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int r = 0; r < 6; ++r) s += Vj[r * 3 + c] * xj[r];
        acc[c] += s;
      }
    }
    for (int c = 0; c < 3; ++c)
      o->y[(size_t)i * 3 + c] = o->Cinvb[(size_t)i * 3 + c] - acc[c];
  }
}

void ba_oracle_backup(ba_oracle *o) {
  o->poses_bak = o->poses;
  o->X_bak = o->X;
}
void ba_oracle_revert(ba_oracle *o) {
  o->poses = o->poses_bak;
  o->X = o->X_bak;
}

// reference :484-500
void ba_oracle_update(ba_oracle *o) {
  for (int j = 0; j < o->N; ++j) {
    Pose &T = o->poses[o->pose_of_j[j]];
    double dR[9], dt[3];
    se3_exp<double>(&o->x[(size_t)6 * j], dR, dt);
    rigid_compose<double>(dR, dt, T.R, T.t, T.R, T.t);
  }
  for (int i = 0; i < o->M; ++i) {
    double *Xi = &o->X[(size_t)o->pt_of_i[i] * 3];
    for (int c = 0; c < 3; ++c) Xi[c] += o->y[(size_t)i * 3 + c];
  }
}

// reference :435-455 (uses the DAMPED A_, C_ — SURVEY Q12)
double ba_oracle_model_change(ba_oracle *o) {
  double est = 0.0;
  for (int j = 0; j < o->N; ++j) {
    const double *xj = &o->x[(size_t)6 * j];
    const double *aj = &o->a[(size_t)6 * j];
    const double *Aj = &o->A[(size_t)36 * j];
    double s = 0.0;
    for (int r = 0; r < 6; ++r) s += aj[r] * xj[r];
    est += s;
    double q = 0.0;
    for (int c = 0; c < 6; ++c) {
      double rowc = 0.0;  // (x^T A)_c
      for (int r = 0; r < 6; ++r) rowc += xj[r] * Aj[r * 6 + c];
      q += rowc * xj[c];
    }
    est += q;
  }
  for (int i = 0; i < o->M; ++i) {
    const double *yi = &o->y[(size_t)3 * i];
    const double *bi = &o->b[(size_t)3 * i];
    const double *Ci = &o->C[(size_t)9 * i];
    est += bi[0] * yi[0] + bi[1] * yi[1] + bi[2] * yi[2];
    double q = 0.0;
    for (int c = 0; c < 3; ++c) {
      const double rowc =
          yi[0] * Ci[0 * 3 + c] + yi[1] * Ci[1 * 3 + c] + yi[2] * Ci[2 * 3 + c];
      q += rowc * yi[c];
    }
    est += q;
    double Bx[3] = {0, 0, 0};
    for (int64_t p = o->lm_ptr[i]; p < o->lm_ptr[i + 1]; ++p) {
      const int j = o->pair_j[p];
      const double *Wj = &o->W[(size_t)p * 18];
      const double *xj = &o->x[(size_t)6 * j];
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int r = 0; r < 6; ++r) s += Wj[r * 3 + c] * xj[r];
        Bx[c] += s;
      }
    }
    est += 2.0 * (yi[0] * Bx[0] + yi[1] * Bx[1] + yi[2] * Bx[2]);
  }
  return -est;
}

void ba_oracle_step_norms(ba_oracle *o, double *pose_sum, double *point_sum) {
  double sp = 0.0, sq = 0.0;
  for (int j = 0; j < o->N; ++j) {
    double s = 0.0;
    for (int r = 0; r < 6; ++r) s += o->x[6 * j + r] * o->x[6 * j + r];
    sp += std::sqrt(s);
  }
  for (int i = 0; i < o->M; ++i) {
    double s = 0.0;
    for (int r = 0; r < 3; ++r) s += o->y[3 * i + r] * o->y[3 * i + r];
    sq += std::sqrt(s);
  }
  *pose_sum = sp;
  *point_sum = sq;
}

// reference :705-1008
int ba_oracle_solve(ba_oracle *o, const ba_oracle_options *opt,
                    ba_oracle_iter *iters, int cap, int *converged) {
  for (int s = 0; s < 4; ++s) o->stage_ms[s] = 0.0;
  const double huber = (double)opt->threshold_huber_loss;
  bool is_converged = false;
  double previous_cost = ba_oracle_cost(o);
  double lambda = (double)opt->initial_lambda;
  int it = 0;
  double t_lap = now_ms();
  for (; it < opt->max_num_iterations; ++it) {
    double t0 = now_ms();
    ba_oracle_linearize(o, huber);
    double t1 = now_ms();
    ba_oracle_damp_invert(o, lambda);
    ba_oracle_schur(o);
    double t2 = now_ms();
    ba_oracle_solve_reduced(o);
    ba_oracle_backsub(o);
    double t3 = now_ms();
    ba_oracle_backup(o);
    ba_oracle_update(o);
    const double current_cost = ba_oracle_cost(o);
    double model = 0.0, rho = 0.0;
    int status = 0;
    if (opt->gauss_newton) {
      // refactor :976-982: update, evaluate the cost, status UPDATE
    } else {
      model = ba_oracle_model_change(o);
      // :930  inverse_scaler_ = 100
      rho = (current_cost - previous_cost) * 100.0 / model;
      if (rho > 0.25) {
        status = 0;
      } else {
        ba_oracle_revert(o);
        status = 2;
      }
      if (rho > 0.5) {
        lambda = std::max(1e-10,
                          (double)(lambda * opt->decrease_ratio_lambda));
        status = 1;
      } else if (rho <= 0.25) {
        lambda = std::min(100.0,
                          (double)(lambda * opt->increase_ratio_lambda));
      }
    }
    const double average_error = current_cost / (double)o->n_obs;
    const double cost_change = std::fabs(current_cost - previous_cost);
    double sp, sq;
    ba_oracle_step_norms(o, &sp, &sq);
    const double avg_step = (sq + sp) / (double)(o->N + o->M);
    if (avg_step < (double)opt->threshold_step_size ||
        cost_change < (double)opt->threshold_cost_change)
      is_converged = true;
    if (it >= opt->max_num_iterations - 1) is_converged = false;
    double t4 = now_ms();
    o->stage_ms[0] += t1 - t0;
    o->stage_ms[1] += t2 - t1;
    o->stage_ms[2] += t3 - t2;
    o->stage_ms[3] += t4 - t3;
    if (iters && it < cap) {
      ba_oracle_iter &I = iters[it];
      I.cost = current_cost;
      I.cost_change = cost_change;
      I.average_reprojection_error = average_error;
      I.abs_step = avg_step;
      I.abs_gradient = 0;
      I.damping_term = lambda;
      I.iter_time_ms = t4 - t_lap;
      I.iteration_status = status;
      I.pad_ = 0;
      I.rho = rho;
      I.model_change = model;
      I.trial_cost = current_cost;
      if (status == 2) {  // :995-1000
        I.cost = previous_cost;
        I.cost_change = 0;
        I.average_reprojection_error =
            std::sqrt(previous_cost / (double)o->n_obs);
      }
    }
    t_lap = t4;
    previous_cost = current_cost;  // :1005 — even when SKIPPED (Q2)
    if (is_converged) {
      ++it;
      break;
    }
  }
  if (converged) *converged = is_converged ? 1 : 0;
  return it;
}

void ba_oracle_stage_ms(const ba_oracle *o, double out4[4]) {
  for (int s = 0; s < 4; ++s) out4[s] = o->stage_ms[s];
}

void ba_oracle_get_poses(const ba_oracle *o, double *T12) {
  for (size_t p = 0; p < o->poses.size(); ++p) {
    std::memcpy(T12 + 12 * p, o->poses[p].R, 9 * sizeof(double));
    std::memcpy(T12 + 12 * p + 9, o->poses[p].t, 3 * sizeof(double));
  }
}
void ba_oracle_get_points(const ba_oracle *o, double *X3) {
  std::memcpy(X3, o->X.data(), o->X.size() * sizeof(double));
}
void ba_oracle_get_A(const ba_oracle *o, double *A36, double *a6) {
  if (A36) std::memcpy(A36, o->A.data(), o->A.size() * sizeof(double));
  if (a6) std::memcpy(a6, o->a.data(), o->a.size() * sizeof(double));
}
void ba_oracle_get_C(const ba_oracle *o, double *C9, double *b3) {
  if (C9) std::memcpy(C9, o->C.data(), o->C.size() * sizeof(double));
  if (b3) std::memcpy(b3, o->b.data(), o->b.size() * sizeof(double));
}
void ba_oracle_get_Cinv(const ba_oracle *o, double *Cinv9, double *Cinvb3) {
  if (Cinv9)
    std::memcpy(Cinv9, o->Cinv.data(), o->Cinv.size() * sizeof(double));
  if (Cinvb3)
    std::memcpy(Cinvb3, o->Cinvb.data(), o->Cinvb.size() * sizeof(double));
}
void ba_oracle_get_pairs(const ba_oracle *o, int32_t *pair_i, int32_t *pair_j,
                         double *W18) {
  for (int i = 0; i < o->M; ++i)
    for (int64_t p = o->lm_ptr[i]; p < o->lm_ptr[i + 1]; ++p) {
      if (pair_i) pair_i[p] = i;
      if (pair_j) pair_j[p] = o->pair_j[p];
    }
  if (W18) std::memcpy(W18, o->W.data(), o->W.size() * sizeof(double));
}
void ba_oracle_get_S(const ba_oracle *o, double *S, double *rhs) {
  if (S) std::memcpy(S, o->S.data(), o->S.size() * sizeof(double));
  if (rhs) std::memcpy(rhs, o->rhs.data(), o->rhs.size() * sizeof(double));
}
void ba_oracle_get_xy(const ba_oracle *o, double *x6, double *y3) {
  if (x6) std::memcpy(x6, o->x.data(), o->x.size() * sizeof(double));
  if (y3) std::memcpy(y3, o->y.data(), o->y.size() * sizeof(double));
}

void ba_oracle_set_S(ba_oracle *o, const double *S, const double *rhs) {
  if (S) std::memcpy(o->S.data(), S, o->S.size() * sizeof(double));
  if (rhs) std::memcpy(o->rhs.data(), rhs, o->rhs.size() * sizeof(double));
}
void ba_oracle_set_x(ba_oracle *o, const double *x6) {
  std::memcpy(o->x.data(), x6, o->x.size() * sizeof(double));
}

void ba_oracle_ldlt_solve(int n, const double *A_rowmajor, int nrhs,
                          const double *B_colmajor, double *X_colmajor) {
  Ldlt f;
  f.compute(n, A_rowmajor);
  for (int c = 0; c < nrhs; ++c) {
    std::memcpy(X_colmajor + (size_t)c * n, B_colmajor + (size_t)c * n,
                n * sizeof(double));
    f.solve(X_colmajor + (size_t)c * n);
  }
}

// ---------------------------------------------------------------------------
// Pose-only monocular 6-DoF, fp32.
// reference core/pose_only_bundle_adjustment_solver.cpp:8-170, helpers
// :1338-1348 (warp), :1350-1384 (Jacobian), :1386-1452 (gradient / Hessian),
// :1147-1200 (upper-triangle append / mirror), :1280-1316 (se3 exp).
// ---------------------------------------------------------------------------
// Shared Gauss-Newton loop of the pose-only solvers.  uvr == nullptr: monocular
// (reference :8-170).  Otherwise stereo (reference :172-399): the right camera
// sees X_r = (left_to_right)^-1 * X_l (Rrl, trl), contributes only where its
// pixel is non-negative (:298), has its own mask, and the error is normalised by
// (count_left + count_right) * 0.5f (:331).
static int pose_only_core(const float *X3, const float *uv2, const float *uvr,
                          int n, float fx, float fy, float cx, float cy,
                          float fxr, float fyr, float cxr, float cyr,
                          const float *Rrl, const float *trl, float *T12,
                          uint8_t *mask, uint8_t *maskr,
                          const ba_oracle_options *opt,
                          ba_oracle_po_iter *iters, int cap, int *n_iter,
                          int *converged, float *debug_T12) {
  const float thr_huber = opt->threshold_huber_loss;
  const float thr_step = opt->threshold_step_size;
  const float thr_cost = opt->threshold_cost_change;
  const float thr_out = opt->threshold_outlier_rejection;
  const int max_it = opt->max_num_iterations;
  const float inv_n = 1.0f / (float)n;
  // pose_camera_to_world_optimized = reference_to_current.inverse()
  // (Isometry inverse: R^T, -R^T t)
  float R[9], t[3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = T12[c * 3 + r];
  for (int r = 0; r < 3; ++r)
    t[r] = -(R[r * 3 + 0] * T12[9] + R[r * 3 + 1] * T12[10] +
             R[r * 3 + 2] * T12[11]);
  bool is_converged = true;
  float err_prev = 1e10f;
  const float lambda = 1e-5f;
  int rows = 0;
  int it = 0;
  for (; it < max_it; ++it) {
    float H[36], g[6];
    for (int k = 0; k < 36; ++k) H[k] = 0.0f;
    for (int k = 0; k < 6; ++k) g[k] = 0.0f;
    float err_curr = 0.0f;
    size_t count_right = 0;
    for (int p = 0; p < n; ++p) {
      const float *Xp = X3 + 3 * p;
      float L[3];
      for (int r = 0; r < 3; ++r)
        L[r] = (R[r * 3 + 0] * Xp[0] + R[r * 3 + 1] * Xp[1] +
                R[r * 3 + 2] * Xp[2]) +
               t[r];
      // one camera's terms (reference :1350-1452)
      auto edge = [&](const float *L, float fx, float fy, float cx, float cy,
                      float pu, float pv, uint8_t *mk) {
      const float iz = 1.0f / L[2];
      const float xiz = L[0] * iz, yiz = L[1] * iz;
      const float fxxiz = fx * xiz, fyyiz = fy * yiz;
      const float ru = (fxxiz + cx) - pu;
      const float rv = (fyyiz + cy) - pv;
      float Ju[6], Jv[6];
      Ju[0] = fx * iz;
      Ju[1] = 0.0f;
      Ju[2] = -fxxiz * iz;
      Ju[3] = -fxxiz * yiz;
      Ju[4] = fx * (1.0f + xiz * xiz);
      Ju[5] = -fx * yiz;
      Jv[0] = 0.0f;
      Jv[1] = fy * iz;
      Jv[2] = -fyyiz * iz;
      Jv[3] = -fy * (1.0f + yiz * yiz);
      Jv[4] = fyyiz * xiz;
      Jv[5] = fy * xiz;
      const float ars = std::fabs(ru) + std::fabs(rv);
      float Hi[36], gi[6];
      for (int k = 0; k < 36; ++k) Hi[k] = 0.0f;
      float error_i = 0.0f;
      if (ars >= thr_huber) {
        const float w = thr_huber / ars;
        const float wru = w * ru, wrv = w * rv;
        for (int r = 0; r < 6; ++r)
          for (int c = r; c < 6; ++c) {
            // x part skips index 1, y part skips index 0 (structural zeros)
            float v = 0.0f;
            if (r != 1 && c != 1) v += (w * Ju[r]) * Ju[c];
            Hi[r * 6 + c] += v;
          }
        for (int r = 0; r < 6; ++r)
          for (int c = r; c < 6; ++c)
            if (r != 0 && c != 0) Hi[r * 6 + c] += (w * Jv[r]) * Jv[c];
        for (int k = 0; k < 6; ++k) gi[k] = wru * Ju[k];
        for (int k = 0; k < 6; ++k) gi[k] += wrv * Jv[k];
        error_i += wru * ru;  // only error_u (Q9)
      } else {
        for (int r = 0; r < 6; ++r)
          for (int c = r; c < 6; ++c)
            if (r != 1 && c != 1) Hi[r * 6 + c] += Ju[r] * Ju[c];
        for (int r = 0; r < 6; ++r)
          for (int c = r; c < 6; ++c)
            if (r != 0 && c != 0) Hi[r * 6 + c] += Jv[r] * Jv[c];
        for (int k = 0; k < 6; ++k) gi[k] = ru * Ju[k];
        for (int k = 0; k < 6; ++k) gi[k] += rv * Jv[k];
        error_i += rv * rv;  // only error_v (Q9)
      }
      for (int k = 0; k < 6; ++k) g[k] -= gi[k];
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) H[r * 6 + c] += Hi[r * 6 + c];
      err_curr += error_i;
      if (ars >= thr_out) mk[p] = 0;
      };
      edge(L, fx, fy, cx, cy, uv2[2 * p + 0], uv2[2 * p + 1], mask);
      if (uvr && !(uvr[2 * p + 0] < 0 || uvr[2 * p + 1] < 0)) {  // :298
        ++count_right;
        float Lr[3];
        for (int r = 0; r < 3; ++r)
          Lr[r] = (Rrl[r * 3 + 0] * L[0] + Rrl[r * 3 + 1] * L[1] +
                   Rrl[r * 3 + 2] * L[2]) +
                  trl[r];
        edge(Lr, fxr, fyr, cxr, cyr, uvr[2 * p + 0], uvr[2 * p + 1], maskr);
      }
    }
    for (int r = 0; r < 6; ++r)
      for (int c = r + 1; c < 6; ++c) H[c * 6 + r] = H[r * 6 + c];
    for (int r = 0; r < 6; ++r) H[r * 6 + r] *= (1.0f + lambda);
    // 6x6 LDLT in fp32: the same pivoted algorithm, evaluated in float
    float d[6];
    {
      // float re-statement of Ldlt for n = 6
      float m[36];
      int tr[6];
      for (int k = 0; k < 36; ++k) m[k] = H[k];
      auto at = [&](int r, int c) -> float & { return m[r * 6 + c]; };
      bool early = false;
      for (int k = 0; k < 6 && !early; ++k) {
        int big = k;
        float bigv = std::fabs(at(k, k));
        for (int i = k + 1; i < 6; ++i)
          if (std::fabs(at(i, i)) > bigv) {
            bigv = std::fabs(at(i, i));
            big = i;
          }
        tr[k] = big;
        if (k != big) {
          const int s = 6 - big - 1;
          for (int c = 0; c < k; ++c) std::swap(at(k, c), at(big, c));
          for (int r = 0; r < s; ++r)
            std::swap(at(big + 1 + r, k), at(big + 1 + r, big));
          std::swap(at(k, k), at(big, big));
          for (int i = k + 1; i < big; ++i) {
            float tt = at(i, k);
            at(i, k) = at(big, i);
            at(big, i) = tt;
          }
        }
        const int rs = 6 - k - 1;
        float tmp[6];
        if (k > 0) {
          float acc = 0.0f;
          for (int c = 0; c < k; ++c) {
            tmp[c] = at(c, c) * at(k, c);
            acc += at(k, c) * tmp[c];
          }
          at(k, k) -= acc;
          for (int r = 0; r < rs; ++r) {
            float s2 = 0.0f;
            for (int c = 0; c < k; ++c) s2 += at(k + 1 + r, c) * tmp[c];
            at(k + 1 + r, k) -= s2;
          }
        }
        const float akk = at(k, k);
        const bool valid = std::fabs(akk) > 0.0f;
        if (k == 0 && !valid) {
          for (int j = 0; j < 6; ++j) tr[j] = j;
          early = true;
          break;
        }
        if (rs > 0 && valid)
          for (int r = 0; r < rs; ++r) at(k + 1 + r, k) /= akk;
      }
      for (int k = 0; k < 6; ++k) d[k] = g[k];
      for (int i = 0; i < 6; ++i)
        if (tr[i] != i) std::swap(d[i], d[tr[i]]);
      for (int i = 0; i < 6; ++i) {
        float s = d[i];
        for (int c = 0; c < i; ++c) s -= at(i, c) * d[c];
        d[i] = s;
      }
      for (int i = 0; i < 6; ++i) {
        if (std::fabs(at(i, i)) > FLT_MIN)
          d[i] /= at(i, i);
        else
          d[i] = 0.0f;
      }
      for (int i = 5; i >= 0; --i) {
        float s = d[i];
        for (int r = i + 1; r < 6; ++r) s -= at(r, i) * d[r];
        d[i] = s;
      }
      for (int i = 5; i >= 0; --i)
        if (tr[i] != i) std::swap(d[i], d[tr[i]]);
    }
    float dR[9], dt[3];
    se3_exp<float>(d, dR, dt);
    rigid_compose<float>(dR, dt, R, t, R, t);
    if (debug_T12 && it < cap) {
      // debug pose = optimized.inverse()
      float *D = debug_T12 + 12 * it;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) D[r * 3 + c] = R[c * 3 + r];
      for (int r = 0; r < 3; ++r)
        D[9 + r] = -(D[r * 3 + 0] * t[0] + D[r * 3 + 1] * t[1] +
                     D[r * 3 + 2] * t[2]);
    }
    if (uvr)
      err_curr /= ((size_t)n + count_right) * 0.5f;  // :331
    else
      err_curr *= (inv_n * 0.5f);
    const float delta_error = std::fabs(err_curr - err_prev);
    float dn = 0.0f;
    for (int k = 0; k < 6; ++k) dn += d[k] * d[k];
    dn = std::sqrt(dn);
    if (dn < thr_step || delta_error < thr_cost) {
      is_converged = true;
      ++it;
      break;  // Q9: no Summary row on the converging iteration
    }
    if (it == max_it - 1) is_converged = false;
    if (iters && rows < cap) {
      iters[rows].cost = err_curr;
      iters[rows].cost_change = delta_error;
      iters[rows].abs_step = dn;
    }
    ++rows;
    err_prev = err_curr;
  }
  if (n_iter) *n_iter = it;
  if (converged) *converged = is_converged ? 1 : 0;
  float nrm = 0.0f;
  for (int k = 0; k < 9; ++k) nrm += R[k] * R[k];
  if (std::isnan(nrm)) return 0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T12[r * 3 + c] = R[c * 3 + r];
  for (int r = 0; r < 3; ++r)
    T12[9 + r] = -(T12[r * 3 + 0] * t[0] + T12[r * 3 + 1] * t[1] +
                   T12[r * 3 + 2] * t[2]);
  return 1;
}


int ba_oracle_pose_only_mono6(const float *X3, const float *uv2, int n,
                              float fx, float fy, float cx, float cy,
                              float *T12, uint8_t *mask,
                              const ba_oracle_options *opt,
                              ba_oracle_po_iter *iters, int cap, int *n_iter,
                              int *converged, float *debug_T12) {
  return pose_only_core(X3, uv2, nullptr, n, fx, fy, cx, cy, 0, 0, 0, 0, nullptr,
                        nullptr, T12, mask, nullptr, opt, iters, cap, n_iter,
                        converged, debug_T12);
}

// reference core/pose_only_bundle_adjustment_solver.cpp:172-399
int ba_oracle_pose_only_stereo6(const float *X3, const float *uvl2,
                                const float *uvr2, int n, const float *intr_l4,
                                const float *intr_r4, const float *T_lr12,
                                float *T12, uint8_t *mask_l, uint8_t *mask_r,
                                const ba_oracle_options *opt,
                                ba_oracle_po_iter *iters, int cap, int *n_iter,
                                int *converged, float *debug_T12) {
  // pose_right_to_left = left_to_right_pose.inverse()  (:226)
  float Rrl[9], trl[3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Rrl[r * 3 + c] = T_lr12[c * 3 + r];
  for (int r = 0; r < 3; ++r)
    trl[r] = -(Rrl[r * 3 + 0] * T_lr12[9] + Rrl[r * 3 + 1] * T_lr12[10] +
               Rrl[r * 3 + 2] * T_lr12[11]);
  return pose_only_core(X3, uvl2, uvr2, n, intr_l4[0], intr_l4[1], intr_l4[2],
                        intr_l4[3], intr_r4[0], intr_r4[1], intr_r4[2], intr_r4[3],
                        Rrl, trl, T12, mask_l, mask_r, opt, iters, cap, n_iter,
                        converged, debug_T12);
}

}  // extern "C"
