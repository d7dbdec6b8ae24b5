// ba_kernels.hip — gfx950 kernels of the full-BA LM iteration.
//
// Stage map (reference core/full_bundle_adjustment_solver.cpp):
//   k_cost                 :381-433   sum of residual norms (stage API; fixed landmarks)
//   k_lin_landmarks        :716-831 (landmark side: C_i, b_i, W_ji) and, as a
//                          by-product, :381-433 the cost at the same parameters
//   k_damp_invert          :846-856 (damp, 3x3 LDLT pseudo-inverse)
//   k_lin_poses/_finalize  :716-810 (pose side: A_j, a_j; damping :833-844 is
//                          applied where A_j is read)
//   k_rhs_partial/_final   :864,:887 rhs_j = a_j - sum_i B_ji (Cinv_i b_i)
//   k_schur_lds/_partial/_final :859-885  S_jk = d_jk A_j - sum_i V_ji W_ki^T,
//                          V_ji = W_ji Cinv_i formed in LDS, never stored
//   k_backsub_update       :910-917 (y_i), :495-499 (X += y), :442-452 model
//   k_pose_update          :487-494 (exp(x) T), :437-441 model, :962 |x|
//   k_scalars / k_control  :928-1007 trust region, convergence, log
//
// All reductions use fixed grids and fixed summation trees: results are
// bitwise reproducible run to run (no floating-point atomics anywhere).
#include "ba_device.h"
#include "ba_plan.h"

namespace ba {

thread_local KernelTimer *g_ktimer = nullptr;

void KernelTimer::begin(int id, hipStream_t s) {
  const size_t k = ids.size();
  while (pool.size() < 2 * (k + 1)) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    pool.push_back(e);
  }
  ids.push_back(id);
  (void)hipEventRecord(pool[2 * k], s);
}
void KernelTimer::end(hipStream_t s) {
  if (ids.empty()) return;
  (void)hipEventRecord(pool[2 * (ids.size() - 1) + 1], s);
}
void KernelTimer::collect() {
  for (size_t k = 0; k < ids.size(); ++k) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, pool[2 * k], pool[2 * k + 1]) == hipSuccess) {
      ms[ids[k]] += t;
      calls[ids[k]] += 1;
    }
  }
  ids.clear();
}
void KernelTimer::reset() {
  for (int k = 0; k < K_COUNT; ++k) {
    ms[k] = 0;
    calls[k] = 0;
  }
  ids.clear();
}

namespace {

constexpr int kBlock = 256;
typedef double v4f64 __attribute__((ext_vector_type(4)));

// Wave-wide sum on the DPP network (no LDS round trips): xor-1, xor-2,
// half-mirror and mirror steps leave every lane with the total of its row of
// 16, the four row totals are then added in order.  Fixed association:
// deterministic; every lane returns the total.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  union { double d; int i[2]; } u, r;
  u.d = v;
  r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, false);
  r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, false);
  return r.d;
}
__device__ __forceinline__ double lane_f64(double v, int src) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return ((lane_f64(v, 0) + lane_f64(v, 16)) + lane_f64(v, 32)) + lane_f64(v, 48);
}

// Sum over a 256-thread block; result valid in thread 0.  `sm` holds >= 4.
__device__ __forceinline__ double block_sum(double v, double *sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) r += sm[w];
  }
  return r;
}

// Two sums over a 256-thread block with one pair of barriers; valid in thread 0.
// `sm` holds >= 8.
__device__ __forceinline__ void block_sum2(double &a, double &b, double *sm) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
    sm[wv] = a;
    sm[4 + wv] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = ((sm[0] + sm[1]) + sm[2]) + sm[3];
    b = ((sm[4] + sm[5]) + sm[6]) + sm[7];
  }
}

// The camera table (16 doubles per camera: fx fy cx cy, R_cj row-major, t_cj)
// is read by every observation: the first kCamLds cameras live in LDS so the
// per-observation gather does not go to the vector memory pipeline.
constexpr int kCamLds = 8;
__device__ __forceinline__ void stage_cams(const DevProblem &d, double *cams_s) {
  const int n = (d.n_cam < kCamLds ? d.n_cam : kCamLds) * 16;
  for (int k = threadIdx.x; k < n; k += blockDim.x) cams_s[k] = d.cams[k];
}
// LDSCAM is a kernel template parameter (chosen at launch: n_cam <= kCamLds),
// not a run-time branch: a divergent fallback inside a pipelined loop makes the
// compiler drain every outstanding load at the join.
template <bool LDSCAM>
__device__ __forceinline__ void load_cam(const DevProblem &d, const double *cams_s,
                                         int idx, double cam[16]) {
  if (LDSCAM) {
#pragma unroll
    for (int k = 0; k < 16; ++k) cam[k] = cams_s[idx * 16 + k];
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) cam[k] = d.cams[(size_t)idx * 16 + k];
  }
}

// Projection of one observation; reference :743-760 / :413-425.
struct ObsGeom {
  double Xij[3];
  double Xc[3];
  double r0, r1;
};

__device__ __forceinline__ void project(const double *__restrict__ cam,
                                        const double *__restrict__ T,
                                        const double X0, const double X1,
                                        const double X2, const double u,
                                        const double v, ObsGeom &g) {
#pragma clang fp contract(fast)
#pragma unroll
  for (int r = 0; r < 3; ++r)
    g.Xij[r] = (T[r * 3 + 0] * X0 + T[r * 3 + 1] * X1 + T[r * 3 + 2] * X2) +
               T[9 + r];
  const double *Rc = cam + 4;
  const double *tc = cam + 13;
#pragma unroll
  for (int r = 0; r < 3; ++r)
    g.Xc[r] = (Rc[r * 3 + 0] * g.Xij[0] + Rc[r * 3 + 1] * g.Xij[1] +
               Rc[r * 3 + 2] * g.Xij[2]) +
              tc[r];
  const double invz = 1.0 / g.Xc[2];
  g.r0 = cam[0] * (g.Xc[0] * invz) + cam[2] - u;
  g.r1 = cam[1] * (g.Xc[1] * invz) + cam[3] - v;
}

// Huber-like weight (reference :763-766) and G = dpi/dXc * R_cj (:770-787).
__device__ __forceinline__ void weight_and_G(const double *__restrict__ cam,
                                             const ObsGeom &g, double huber,
                                             double &w, double G[6]) {
#pragma clang fp contract(fast)
  const double invz = 1.0 / g.Xc[2];
  const double fxinvz = cam[0] * invz, fyinvz = cam[1] * invz;
  const double xinvz = g.Xc[0] * invz, yinvz = g.Xc[1] * invz;
  const double fx_xinvz2 = fxinvz * xinvz, fy_yinvz2 = fyinvz * yinvz;
  const double absr = fabs(g.r0) + fabs(g.r1);
  w = (absr > huber) ? (huber / absr) : 1.0;
  const double *Rc = cam + 4;
  G[0] = fxinvz * Rc[0] + (-fx_xinvz2) * Rc[6];
  G[1] = fxinvz * Rc[1] + (-fx_xinvz2) * Rc[7];
  G[2] = fxinvz * Rc[2] + (-fx_xinvz2) * Rc[8];
  G[3] = fyinvz * Rc[3] + (-fy_yinvz2) * Rc[6];
  G[4] = fyinvz * Rc[4] + (-fy_yinvz2) * Rc[7];
  G[5] = fyinvz * Rc[5] + (-fy_yinvz2) * Rc[8];
}

// Q = [G, G * (-[Xij]x)]  (reference :797-800), 2x6 row-major
__device__ __forceinline__ void make_Q(const double G[6], const double Xij[3],
                                       double Q[12]) {
#pragma clang fp contract(fast)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const double g0 = G[r * 3 + 0], g1 = G[r * 3 + 1], g2 = G[r * 3 + 2];
    Q[r * 6 + 0] = g0;
    Q[r * 6 + 1] = g1;
    Q[r * 6 + 2] = g2;
    Q[r * 6 + 3] = g2 * Xij[1] - g1 * Xij[2];
    Q[r * 6 + 4] = g0 * Xij[2] - g2 * Xij[0];
    Q[r * 6 + 5] = g1 * Xij[0] - g0 * Xij[1];
  }
}

// R = G * R_jw (reference :814), 2x3 row-major
__device__ __forceinline__ void make_R(const double G[6],
                                       const double *__restrict__ T,
                                       double Rm[6]) {
#pragma clang fp contract(fast)
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      Rm[r * 3 + c] = G[r * 3 + 0] * T[0 * 3 + c] + G[r * 3 + 1] * T[1 * 3 + c] +
                      G[r * 3 + 2] * T[2 * 3 + c];
}

// rows 3..5 of B_ji from its compact record {K (9), X_ij (3)}: row 3+a, column c
// = (X_ij x K[:,c])[a]
__device__ __forceinline__ void expand_W(const double *__restrict__ k12, double W[18]) {
#pragma unroll
  for (int e = 0; e < 9; ++e) W[e] = k12[e];
  const double X0 = k12[9], X1 = k12[10], X2 = k12[11];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    W[9 + c] = X1 * k12[6 + c] - X2 * k12[3 + c];
    W[12 + c] = X2 * k12[c] - X0 * k12[6 + c];
    W[15 + c] = X0 * k12[3 + c] - X1 * k12[c];
  }
}

// Symmetric 3x3 inverse by diagonally pivoted LDL^T with D pseudo-inverted —
// the behaviour of Eigen's C.ldlt().solve(I) (reference :854): an all-zero
// C_i (never-observed landmark) yields Cinv = 0, not NaN.
// c = {c00 c01 c02 c11 c12 c22}; out in the same order.
__device__ __forceinline__ void ldlt3_inverse(const double c[6], double o[6]) {
  double a00 = c[0], a01 = c[1], a02 = c[2], a11 = c[3], a12 = c[4], a22 = c[5];
  // pivot order = selection by |diag| (left-looking: untouched diagonal)
  int s0 = 0;  // 0: none, 1: swap(0,1), 2: swap(0,2)
  {
    double m = fabs(a00);
    if (fabs(a11) > m) {
      m = fabs(a11);
      s0 = 1;
    }
    if (fabs(a22) > m) s0 = 2;
  }
  double t;
  if (s0 == 1) {
    t = a00; a00 = a11; a11 = t;
    t = a02; a02 = a12; a12 = t;
  } else if (s0 == 2) {
    t = a00; a00 = a22; a22 = t;
    t = a01; a01 = a12; a12 = t;
  }
  const bool s1 = fabs(a22) > fabs(a11);
  if (s1) {
    t = a11; a11 = a22; a22 = t;
    t = a01; a01 = a02; a02 = t;
  }
  double b00 = 0, b01 = 0, b02 = 0, b11 = 0, b12 = 0, b22 = 0;
  const double d0 = a00;
  if (fabs(d0) > 0.0) {
    const double l10 = a01 / d0, l20 = a02 / d0;
    const double tmp0 = d0 * l10;
    const double d1 = a11 - l10 * tmp0;
    double l21 = a12 - l20 * tmp0;
    if (fabs(d1) > 0.0) l21 /= d1;
    const double d2 = a22 - (l20 * (d0 * l20) + l21 * (d1 * l21));
    const double tol = 2.2250738585072014e-308;
    const double i0 = (fabs(d0) > tol) ? 1.0 / d0 : 0.0;
    const double i1 = (fabs(d1) > tol) ? 1.0 / d1 : 0.0;
    const double i2 = (fabs(d2) > tol) ? 1.0 / d2 : 0.0;
    // columns of the inverse: solve L D L^T x = e_c
    // e0: z = (1, -l10, -l20 + l21 l10)
    {
      const double z0 = 1.0, z1 = -l10 * z0, z2 = -l20 * z0 - l21 * z1;
      const double w0 = z0 * i0, w1 = z1 * i1, w2 = z2 * i2;
      const double x2 = w2, x1 = w1 - l21 * x2, x0 = w0 - l10 * x1 - l20 * x2;
      b00 = x0;
      (void)x1;
      (void)x2;
    }
    {
      const double z1 = 1.0, z2 = -l21 * z1;
      const double w1 = z1 * i1, w2 = z2 * i2;
      const double x2 = w2, x1 = w1 - l21 * x2, x0 = -l10 * x1 - l20 * x2;
      b01 = x0;
      b11 = x1;
    }
    {
      const double w2 = i2;
      const double x2 = w2, x1 = -l21 * x2, x0 = -l10 * x1 - l20 * x2;
      b02 = x0;
      b12 = x1;
      b22 = x2;
    }
  }
  // undo the symmetric permutations (involutions, reverse order)
  if (s1) {
    t = b11; b11 = b22; b22 = t;
    t = b01; b01 = b02; b02 = t;
  }
  if (s0 == 1) {
    t = b00; b00 = b11; b11 = t;
    t = b02; b02 = b12; b12 = t;
  } else if (s0 == 2) {
    t = b00; b00 = b22; b22 = t;
    t = b01; b01 = b12; b12 = t;
  }
  o[0] = b00; o[1] = b01; o[2] = b02; o[3] = b11; o[4] = b12; o[5] = b22;
}

// The same inverse on its fast path: C_i with a damped diagonal is symmetric
// positive definite for every observed landmark, and LDL^T without pivoting is
// backward stable for such a matrix: three reciprocals (v_rcp_f64 + two Newton
// steps instead of IEEE divisions) and no pivot search / permutation selects — a
// third of the instructions of the pivoted routine, which remains the fallback
// whenever a pivot is not safely positive (below 1e-6 of the largest diagonal entry:
// cond(C_i) > ~1e6): the degenerate cases (never-observed landmark, rank-deficient
// C_i) keep Eigen's pseudo-inverse semantics exactly, and an ill-conditioned C_i — a
// landmark whose depth is barely observable, the ones the LM loop lets run away once
// lambda has fallen — is inverted with the reference's own pivot order (multipliers
// <= 1: no overflow of the inverse where the unpivoted order has |l21| ~ 1e3).
__device__ __forceinline__ double rcp_newton(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ void spd3_inverse(const double c[6], double o[6]) {
  const double a00 = c[0], a01 = c[1], a02 = c[2], a11 = c[3], a12 = c[4], a22 = c[5];
  const double thr = 1e-6 * fmax(a00, fmax(a11, a22));
  const double i0 = rcp_newton(a00);
  const double l10 = a01 * i0, l20 = a02 * i0;
  const double d1 = fma(-l10, a01, a11);
  const double t21 = fma(-l20, a01, a12);
  const double i1 = rcp_newton(d1);
  const double l21 = t21 * i1;
  const double d2 = fma(-l21, t21, fma(-l20, a02, a22));
  const double i2 = rcp_newton(d2);
  if (!(a00 > thr && d1 > thr && d2 > thr) || !(thr > 0.0)) {  // (also NaN / zero matrices)
    ldlt3_inverse(c, o);
    return;
  }
  // inverse = L^-T D^-1 L^-1 with L^-1 = [1 0 0; -l10 1 0; l10 l21 - l20, -l21, 1]
  const double m20 = fma(l10, l21, -l20);
  const double w2 = m20 * i2, v2 = l21 * i2;
  o[5] = i2;                                  // (2,2)
  o[4] = -v2;                                 // (1,2)
  o[2] = w2;                                  // (0,2)
  o[3] = fma(l21, v2, i1);                    // (1,1) = i1 + l21^2 i2
  o[1] = fma(-l10, i1, -(l21 * w2));          // (0,1) = -l10 i1 - l21 m20 i2
  o[0] = fma(m20, w2, fma(l10 * l10, i1, i0));  // (0,0) = i0 + l10^2 i1 + m20^2 i2
}

// Observation record as the cost kernel reads it: {camera, pose, point}.  SLIM:
// the 8-byte copy obs_cp = {camera | pose << 16, point} (problems with fewer
// than 65 536 cameras and poses) instead of the 16-byte record that also carries
// the pair id.  The fields are decoded where they are USED, one step after the
// load: arithmetic on a just-loaded prefetch value would wait for it.
template <bool SLIM>
struct ObsRec;
template <>
struct ObsRec<true> {
  int2 v;
  __device__ __forceinline__ int cam() const { return v.x & 0xffff; }
  __device__ __forceinline__ int pose() const { return (int)((unsigned)v.x >> 16); }
  __device__ __forceinline__ int pt() const { return v.y; }
  __device__ __forceinline__ void load(const DevProblem &d, int64_t s) { v = d.obs_cp[s]; }
  __device__ __forceinline__ void clear() { v = make_int2(0, 0); }
};
template <>
struct ObsRec<false> {
  int4 v;
  __device__ __forceinline__ int cam() const { return v.x; }
  __device__ __forceinline__ int pose() const { return v.y; }
  __device__ __forceinline__ int pt() const { return v.z; }
  __device__ __forceinline__ void load(const DevProblem &d, int64_t s) { v = d.obs_idx[s]; }
  __device__ __forceinline__ void clear() { v = make_int4(0, 0, 0, 0); }
};

// --------------------------------------------------------------------------
// cost: sum over observations of ||r||  (reference :381-433)
// --------------------------------------------------------------------------
// Grid-stride over observations, software-pipelined: while observation s is
// projected, the gathered pose / point of s+stride and the record of
// s+2*stride are in flight.  The pose/point registers ping-pong (loop unrolled
// by two) and the record registers are reloaded right after their last use, so
// no register move ever has to wait for a load issued in the same step.
#define COST_STEP(TC, XC, TN, XN)                                               \
  {                                                                             \
    /* gathers of the next observation (record arrived one step ago) */         \
    const double *Tp_ = poses + (size_t)idn.pose() * 12;                        \
    const double *Xp_ = pts + (size_t)idn.pt() * 3;                             \
    _Pragma("unroll") for (int k_ = 0; k_ < 12; ++k_) TN[k_] = Tp_[k_];         \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) XN[k_] = Xp_[k_];          \
    const int ncam_ = idn.cam();                                                \
    const double2 nuv_ = uvn;                                                   \
    /* record two strides ahead, into the registers just read (clamped, not  */ \
    /* conditional: a join after a conditional load waits for it)            */ \
    {                                                                           \
      const int64_t s2_ = s + 2 * stride < d.n_obs ? s + 2 * stride : d.n_obs - 1; \
      idn.load(d, s2_);                                                         \
      uvn = d.obs_uv[s2_];                                                      \
    }                                                                           \
    ObsGeom g_;                                                                 \
    /* camera through a pointer (LDS table or global): fewer live registers */  \
    const double *cam_ = LDSCAM ? (const double *)(cams_s + ccam * 16)          \
                                : (const double *)(d.cams + (size_t)ccam * 16); \
    project(cam_, TC, XC[0], XC[1], XC[2], cuv.x, cuv.y, g_);                   \
    /* (a padded slot of a masked covisibility group has uv = NaN: no observation) */ \
    acc += cuv.x == cuv.x ? sqrt(g_.r0 * g_.r0 + g_.r1 * g_.r1) : 0.0;          \
    ccam = ncam_;                                                               \
    cuv = nuv_;                                                                 \
  }

template <bool LDSCAM, bool SLIM>
__global__ __launch_bounds__(kBlock) void k_cost(DevProblem d, int sel, int64_t begin) {
  __shared__ double sm[4];
  __shared__ double cams_s[kCamLds * 16];
  if (LDSCAM) stage_cams(d, cams_s);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t s = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x;
  // first record issued before the control word is needed
  ObsRec<SLIM> id0;
  id0.clear();
  double2 cuv = make_double2(0.0, 0.0);
  if (s < d.n_obs) {
    id0.load(d, s);
    cuv = d.obs_uv[s];
  }
  if (d.ctrl->done) return;
  // sel 1 inside the LM loop: the trial buffer recorded by k_pose_update; the stage
  // API (ba_stage_scalars) asks for "the other buffer" of the host-set cur
  const int buf = sel == 2 ? d.ctrl->tcur : (d.ctrl->cur ^ sel);
  const double *__restrict__ poses = d.poses[buf];
  const double *__restrict__ pts = d.pts[buf];
  // past-the-end prefetches are clamped to the last record: harmless gathers
  const int64_t s1 = s + stride < d.n_obs ? s + stride : d.n_obs - 1;
  ObsRec<SLIM> idn;
  idn.load(d, s1);
  double2 uvn = d.obs_uv[s1];
  double TA[12], XA[3], TB[12], XB[3];
  {
    const double *Tp = poses + (size_t)id0.pose() * 12;
    const double *Xp = pts + (size_t)id0.pt() * 3;
#pragma unroll
    for (int k = 0; k < 12; ++k) TA[k] = Tp[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) XA[k] = Xp[k];
  }
  int ccam = id0.cam();
  double acc = 0.0;
  __syncthreads();  // cams_s
  while (s < d.n_obs) {
    COST_STEP(TA, XA, TB, XB)
    s += stride;
    if (s >= d.n_obs) break;
    COST_STEP(TB, XB, TA, XA)
    s += stride;
  }
  const double tot = block_sum(acc, sm);
  if (threadIdx.x == 0) d.cost_part[blockIdx.x] = tot;
}

// --------------------------------------------------------------------------
// landmark side of the linearisation (reference :716-831, :846-856).
// One workgroup per chunk of consecutive landmarks (the chunks of the
// back-substitution: <= kSchurLandmarks landmarks, <= kSchurPairs pairs).
// Phase A, one thread per OBSERVATION (coalesced 16-byte record loads):
//   residual, weight, G, R = G R_jw; its C_i / b_i contribution (9 doubles) goes
//   to LDS, its cross block W_ji = w Q^T R (if it is the pair's last writer,
//   reference :826) to an LDS image of the chunk's W range.
// Phase B, one thread per LANDMARK: sums its observations' contributions in
//   insertion order (deterministic), damps, inverts (pivoted 3x3 LDL^T with
//   pseudo-inverse) and stores Cd, b, Cinv, Cinv b.
// The W image is written back with contiguous 16-byte stores.
// --------------------------------------------------------------------------
#ifdef BA_LL_DBG
__device__ long long g_ll_dbg[32];
#define LL_STAMP() { if (ll_on && ll_n < 32) ll_s[ll_n++] = clock64(); }
#else
#define LL_STAMP()
#endif
template <bool LDSCAM>
__global__ __launch_bounds__(kBlock, 5) void k_lin_landmarks(DevProblem d, int sel) {
#ifdef BA_LL_DBG
  __shared__ long long ll_s[32];
  const bool ll_on = blockIdx.x == 9000 && threadIdx.x == 0;
  int ll_n = 0;
#endif
  LL_STAMP()
  __shared__ __attribute__((aligned(16))) double Wst[kSchurPairs * kWStride];
  __shared__ double Cb[kBlock * 9];
  __shared__ double cams_s[kCamLds * 16];
  __shared__ int lq[kSchurLandmarks + 1];  // landmark observation offsets in the chunk
  __shared__ double smc[4];
  // dependent-load chain: chunk record (+ control word) -> observation records
  // (+ this thread's landmark range) -> pose / point gathers
  // (the chunks before lin_chunk0 hold the covisibility groups: k_lin_grp's)
  const int chunk = blockIdx.x + d.lin_chunk0;
  const DevProblem::LmChunk lc = d.lm_chunk[chunk];
  const int done = d.ctrl->done;
  const int buf = sel ? d.ctrl->tcur : d.ctrl->cur;
  const int lb = sel ? d.ctrl->tlcur : d.ctrl->lcur;
  const double huber = d.ctrl->huber;
  BA_KEEP_S((int)lc.pb);
  BA_KEEP_S((int)lc.ob);
  BA_KEEP_S(lc.l0);
  BA_KEEP_S(lc.no);
  BA_KEEP_S(done);
  if (LDSCAM) stage_cams(d, cams_s);
  const int tid = threadIdx.x;
  const int64_t pb = lc.pb;
  const int npair = lc.np;
  const int64_t ob = lc.ob, oe = lc.ob + lc.no;
  int4 id = make_int4(0, 0, 0, -1);
  double2 uv = make_double2(0.0, 0.0);
  if (tid < lc.no) {
    id = d.obs_idx[ob + tid];
    uv = d.obs_uv[ob + tid];
  }
  const int i = lc.l0 + tid;
  const bool own = tid < lc.nl;
  int64_t q0 = 0;
  if (own) q0 = d.lm_obs_ptr[i];
  if (done) return;
  LL_STAMP()
  const double *__restrict__ poses = d.poses[buf];
  const double *__restrict__ pts = d.pts[buf];
  double *__restrict__ Wg = d.W[lb];
  double cost_acc = 0.0;  // sum of ||r|| over this thread's observations (reference :381-433)
  // landmark observation ranges in LDS; running sums of thread (landmark, value)
  if (tid <= lc.nl) lq[tid] = (tid < lc.nl) ? (int)(q0 - ob) : lc.no;
  constexpr int kSumIt = (kSchurLandmarks * 9 + kBlock - 1) / kBlock;
  double csum[kSumIt];
#pragma unroll
  for (int k = 0; k < kSumIt; ++k) csum[k] = 0.0;
  __syncthreads();  // cams_s, lq
  for (int64_t t0 = ob; t0 < oe; t0 += kBlock) {
    const int64_t s = t0 + tid;
    if (t0 > ob && s < oe) {  // further tiles (more than kBlock observations)
      id = d.obs_idx[s];
      uv = d.obs_uv[s];
    }
    if (s < oe) {
#pragma clang fp contract(fast)
      double cam[16];
      load_cam<LDSCAM>(d, cams_s, id.x, cam);
      const double *T = poses + (size_t)id.y * 12;
      const double *X = pts + (size_t)id.z * 3;
      ObsGeom g;
      project(cam, T, X[0], X[1], X[2], uv.x, uv.y, g);
      cost_acc += sqrt(g.r0 * g.r0 + g.r1 * g.r1);
      double w, G[6], Rm[6];
      weight_and_G(cam, g, huber, w, G);
      make_R(G, T, Rm);
      // reference :503-517, :817-823
      double *cb = Cb + tid * 9;
      cb[0] = w * (Rm[0] * Rm[0] + Rm[3] * Rm[3]);
      cb[1] = w * (Rm[0] * Rm[1] + Rm[3] * Rm[4]);
      cb[2] = w * (Rm[0] * Rm[2] + Rm[3] * Rm[5]);
      cb[3] = w * (Rm[1] * Rm[1] + Rm[4] * Rm[4]);
      cb[4] = w * (Rm[1] * Rm[2] + Rm[4] * Rm[5]);
      cb[5] = w * (Rm[2] * Rm[2] + Rm[5] * Rm[5]);
      const double wr0 = w * g.r0, wr1 = w * g.r1;
      cb[6] = Rm[0] * wr0 + Rm[3] * wr1;
      cb[7] = Rm[1] * wr0 + Rm[4] * wr1;
      cb[8] = Rm[2] * wr0 + Rm[5] * wr1;
      if (id.w >= 0) {
        // B_ji = w Q^T R, kept only from the last-inserted observation of the
        // pair (reference :826), stored compact: K = w G^T R (its rows
        // 0..2) and X_ij (ba_device.h kWStride)
        const int lp = (int)(id.w - pb);
        double *Wp = (lp < kSchurPairs) ? (Wst + lp * kWStride)
                                        : (Wg + (size_t)id.w * kWStride);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c)
            Wp[r * 3 + c] = w * (G[r] * Rm[c] + G[3 + r] * Rm[3 + c]);
#pragma unroll
        for (int r = 0; r < 3; ++r) Wp[9 + r] = g.Xij[r];
      }
    }
    LL_STAMP()
    __syncthreads();
    LL_STAMP()
    // per-landmark sums of this tile, one thread per (landmark, value): nine
    // times more lanes than one thread per landmark, same insertion order
    {
      const int o0 = (int)(t0 - ob);
#pragma unroll
      for (int k = 0; k < kSumIt; ++k) {
        const int t = tid + k * kBlock;
        if (t < lc.nl * 9) {
          const int li = t / 9, v = t - li * 9;
          const int a = max(lq[li], o0), b = min(lq[li + 1], o0 + kBlock);
          double sacc = 0.0;
          // same insertion order, four LDS reads requested before the first add
          const double *cp = Cb + (a - o0) * 9 + v;
          int n = b - a;
          for (; n >= 4; n -= 4, cp += 36) {
            const double v0 = cp[0], v1 = cp[9], v2 = cp[18], v3 = cp[27];
            sacc = (((sacc + v0) + v1) + v2) + v3;
          }
          for (; n > 0; --n, cp += 9) sacc += cp[0];
          csum[k] += sacc;
        }
      }
    }
    LL_STAMP()
    __syncthreads();
    LL_STAMP()
  }
  // sums -> LDS (Cb is free now) -> the landmark's owner thread
#pragma unroll
  for (int k = 0; k < kSumIt; ++k) {
    const int t = tid + k * kBlock;
    if (t < lc.nl * 9) Cb[t] = csum[k];
  }
  __syncthreads();
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
  double b0 = 0, b1 = 0, b2 = 0;
  if (own) {
    const double *cs = Cb + tid * 9;
    c00 = cs[0]; c01 = cs[1]; c02 = cs[2];
    c11 = cs[3]; c12 = cs[4]; c22 = cs[5];
    b0 = -cs[6]; b1 = -cs[7]; b2 = -cs[8];
  }
  // W image -> global (contiguous)
  {
    const int n2 = min(npair, kSchurPairs) * (kWStride / 2);
    const double2 *src = (const double2 *)Wst;
    double2 *dst = (double2 *)(Wg + (size_t)pb * kWStride);
    constexpr int kIt = (kSchurPairs * (kWStride / 2) + kBlock - 1) / kBlock;
    double2 wv[kIt];
#pragma unroll
    for (int k = 0; k < kIt; ++k) {
      const int t = tid + k * kBlock;
      wv[k] = src[t < n2 ? t : 0];
    }
#pragma unroll
    for (int k = 0; k < kIt; ++k) {
      const int t = tid + k * kBlock;
      if (t < n2) dst[t] = wv[k];
    }
  }
  LL_STAMP()
#ifdef BA_LL_DBG
  if (ll_on) { for (int q = 0; q < 32; ++q) g_ll_dbg[q] = q < ll_n ? ll_s[q] : 0; }
#endif
  // cost partial of this workgroup (k_scalars adds them in block order)
  {
    const double tot = block_sum(cost_acc, smc);
    if (tid == 0) d.lin_cost_part[chunk] = tot;
  }
  if (!own) return;
  // undamped C_i (upper) and b_i; damping and the inverse follow the control step
  // (k_damp_invert), because lambda is not known yet when the trial point is
  // linearised
  double *Co = d.Cu[lb] + (size_t)i * 6;
  Co[0] = c00; Co[1] = c01; Co[2] = c02; Co[3] = c11; Co[4] = c12; Co[5] = c22;
  double *bo = d.b[lb] + (size_t)i * 3;
  bo[0] = b0;
  bo[1] = b1;
  bo[2] = b2;
}

// --------------------------------------------------------------------------
// Linearisation of the covisibility groups, landmark AND pose side in one pass
// (reference :716-856).  The landmarks of a group share one observation pattern
// (ba_plan.cpp): observation oo of every landmark is made by the same pose
// through the same camera.  One workgroup per group piece, its four waves
// independent; a wave step covers nlw = 64 / no consecutive landmarks, lane =
// (landmark, pattern slot).  Pose and camera are LANE CONSTANTS (registers for
// the whole kernel), the observation stream is read contiguously, the point is a
// broadcast load: no index records, no gathers.  Per step:
//   residual, weight, G, R, Q;  C_i / b_i terms through a wave-private LDS
//   transposition to one lane per (landmark, value) that adds them in insertion
//   order;  the compact cross block {K, X_ij} from the pair's last writer;
//   A_j / a_j terms (21 + 6) into per-lane register accumulators.
// At the end the accumulators are summed per pose of the group (over waves,
// landmarks of a step and the pattern slots of that pose, fixed order) into one
// row of Apart2 per (group piece, pose); k_pose_finalize adds the rows of a pose.
// --------------------------------------------------------------------------
#ifdef BA_LG_DBG
__device__ long long g_lg_dbg[64];
#define LG_STAMP() { if (lg_on && lg_n < 64) lg_s[lg_n++] = clock64(); }
#else
#define LG_STAMP()
#endif
// MASKED: pieces of superset groups (ba_plan.h GrpRange::masked).  A slot whose uv is
// NaN is a padded one — the landmark has no observation there: weight 0, residual 0,
// no cost — and the pair's last WRITER (reference :826) is the last VALID slot of its
// pose in this landmark, found from a ballot of the step's valid lanes; when a pose has
// no valid slot at all its static last slot stores the (zero-K) record, so that the W
// image never keeps a previous step's entry.
// FUSED (k_backsub_lin: the piece's landmarks were back-substituted and the poses updated
// by workgroups of the SAME launch): the trial buffers are the other ones of ctrl->cur /
// lcur, the workgroup waits for the pose workgroups' counter and for its piece's flag
// (bounded polls) and reads poses and points with sc1 loads.
// per wave: step transposition [64][9] + W image of the step; at the end [14][64]
constexpr int kLgArea = 64 * 9 + 7 * 10 * 12;
template <bool LDSCAM, bool MASKED, bool FUSED>
__device__ __forceinline__ void lin_grp_body(const DevProblem &d, const int sel, const int bid, double *red,
                                             double *cams_s, double *smc, int *slot_b, int *slot_e,
                                             const int *bl_flag = nullptr, const int *pose_done = nullptr,
                                             int gen = 0, int *bad = nullptr) {
#ifdef BA_LG_DBG
  __shared__ long long lg_s[64];
  const bool lg_on = blockIdx.x == 300 && threadIdx.x == 0;
  int lg_n = 0;
#endif
  LG_STAMP()
  const DevProblem::LinDesc *gp = d.lin_desc + bid;
  const int64_t p0 = gp->p0, o0 = gp->o0;
  const int l0 = gp->l0, nl = gp->nl, dd = gp->d, no = gp->no, pat0 = gp->pat0;
  const int apart0 = gp->apart0, cost_idx = gp->cost_idx;
  const int done = d.ctrl->done;
  const int buf = FUSED ? (d.ctrl->cur ^ 1) : (sel ? d.ctrl->tcur : d.ctrl->cur);
  const int lb = FUSED ? (d.ctrl->lcur ^ 1) : (sel ? d.ctrl->tlcur : d.ctrl->lcur);
  const double huber = d.ctrl->huber;
  if (LDSCAM) stage_cams(d, cams_s);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (a scalar: everything derived from it leaves the vector registers)
  // landmarks per wave step; at most 7, so that the 9 nlw (landmark, value) sums of a
  // step are ONE lane each
  const int nlw = 64 / no < 7 ? 64 / no : 7;
  const int ilw = lane / no, oo = lane - ilw * no;
  const bool lane_on = ilw < nlw;
  const int2 pat = d.grp_pat[pat0 + oo];
  if (done) return;
  const int cam_id = pat.y & 0xffff, jj = (pat.y >> 16) & 0xff;
  const bool opt = (pat.y >> 29) & 1, lastw = (pat.y >> 30) & 1;
  // pattern slots [slot_b, slot_e) of each optimisable pose (the slots of a pose are
  // adjacent): found from the neighbours' pattern entries, no further loads
  {
    const int pose_prev = __shfl_up(pat.x, 1), pose_next = __shfl_down(pat.x, 1);
    if (tid < no && opt) {
      if (oo == 0 || pose_prev != pat.x) slot_b[jj] = oo;
      if (oo == no - 1 || pose_next != pat.x) slot_e[jj] = oo + 1;
    }
  }
  const double *__restrict__ pts = d.pts[buf];
#define LING_LDX(p_) (*(p_))
  if (FUSED) {
    if (tid <= kPoseGrid) {  // lanes 0..kPoseGrid-1: the pose workgroups' flags; lane kPoseGrid: the piece's flag
      const int *fl = tid < kPoseGrid ? pose_done + tid : bl_flag + bid;
      int spins = 0;
      while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 22)) {  // ~ a second: give up, flag the iteration as failed
          if (bad) atomicAdd(bad, ::ba::kFlowTimeout);
          break;
        }
      }
    }
    __syncthreads();
    // the piece's points (<= 672 landmarks): ONE sc1 read into the LDS behind the wave areas,
    // the steps then take them from there (sc1 loads inside the step loop — past the L2,
    // two steps of prefetch distance — made the role 2.4 times slower)
    double *Xs = red + 4 * kLgArea;
    for (int e = tid; e < nl * 3; e += kBlock)
      Xs[e] = __hip_atomic_load(&pts[(size_t)l0 * 3 + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pts = Xs - (size_t)l0 * 3;  // (the barrier below also covers Xs)
  }
  double T[12], cam[16];
  {
    const double *Tp = d.poses[buf] + (size_t)pat.x * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k)
      T[k] = FUSED ? __hip_atomic_load(&Tp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : Tp[k];
  }
  __syncthreads();  // cams_s, slot_b
  load_cam<LDSCAM>(d, cams_s, cam_id, cam);
  // MASKED: this lane's slot range of its pose inside the landmark's pattern word
  // (two lane constants: the bits of the LATER slots of this lane's pose and the bits of
  //  all slots of the pose, positions inside the landmark's no-bit word)
  unsigned later_bits = 0u, pose_bits = 0u;
  if (MASKED && opt) {
    const int sb_l = slot_b[jj], se_l = slot_e[jj];
    const unsigned upto_e = se_l >= 32 ? 0xffffffffu : ((1u << se_l) - 1u);
    pose_bits = upto_e & ~((1u << sb_l) - 1u);
    later_bits = oo + 1 >= 32 ? 0u : (upto_e & ~((1u << (oo + 1)) - 1u));
  }
  double *__restrict__ Wg = d.W[lb];
  double *__restrict__ Cg = d.Cu[lb];
  double *__restrict__ bg = d.b[lb];
  double *cbw = red + wv * kLgArea;
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  double cost_acc = 0.0;
  const int per_step = 4 * nlw;
  const int nstep = (nl + per_step - 1) / per_step;
  // the observation stream and the points of the next two steps are in flight
  // while a step is processed (clamped indices: no conditional loads)
  auto il_of = [&](int st) { return (st * 4 + wv) * nlw + ilw; };
  auto clampi = [&](int il) { return min(il, nl - 1); };
  double2 uvq[3];
  double Xq[3][3];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ilc = clampi(il_of(u));
    uvq[u] = d.obs_uv[o0 + (int64_t)ilc * no + oo];
    const double *Xp = pts + (size_t)(l0 + ilc) * 3;
    Xq[u][0] = LING_LDX(&Xp[0]);
    Xq[u][1] = LING_LDX(&Xp[1]);
    Xq[u][2] = LING_LDX(&Xp[2]);
  }
  // this lane's (landmark, value) sum of a wave step and where it goes
  const int sum_li = lane / 9, sum_v = lane - sum_li * 9;
  const bool sum_on = lane < nlw * 9, sum_neg = sum_v >= 6;
  const double *sum_p = cbw + (sum_on ? sum_li * no * 9 + sum_v : 0);
  double *sum_g = sum_v < 6 ? Cg + (size_t)l0 * 6 + sum_v : bg + (size_t)l0 * 3 + (sum_v - 6);
  const int sum_stride = sum_v < 6 ? 6 : 3;
  // dump area of the lanes that have nothing to store (see LING_STEP)
  double2 *dumpW = (double2 *)(d.lin_dump + (size_t)((bid & 63) * 4 + wv) * 4);
  double *dumpC = (double *)dumpW + 2;
  // LDS image of the W records of a wave step (behind the transposition area):
  // nlw * d <= 7 * 10 records of 6 double2
  double2 *stgW = (double2 *)(cbw + 64 * 9);
  double2 *stgW_mine = stgW + (ilw * dd + jj) * 6;
  constexpr int kLgPass = (7 * 10 * 6 + 63) / 64;
  // A step is software-pipelined against the previous one: the step's terms go to
  // the wave's LDS area at its END; the per-landmark sums and the W image of step
  // st - 1 are READ from LDS at the start of step st (requests in flight during the
  // arithmetic of step st) and stored to global after it.  il0p = first landmark of
  // the step whose terms are in LDS (nl: none yet, everything goes to the dump).
  int il0p = nl;
  double vv[8];
  double2 wvv[kLgPass];
  int lastp = 0;
  double2 *dstWp = dumpW;
  // requests of the LDS reads for the step in LDS
#define LING_FLUSH_ISSUE()                                                          \
  {                                                                                 \
    _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) vv[u_] = sum_p[(u_ < no ? u_ : 0) * 9]; \
    const int nls_ = nl - il0p < nlw ? nl - il0p : nlw;                             \
    const int n2_ = nls_ * dd * 6;                                                  \
    dstWp = n2_ > 0 ? (double2 *)(Wg + (size_t)(p0 + (int64_t)il0p * dd) * kWStride) : dumpW; \
    lastp = n2_ > 0 ? n2_ - 1 : 0;                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < kLgPass; ++k_) wvv[k_] = stgW[min(lane + 64 * k_, lastp)]; \
  }
  // per-landmark sums in insertion order (one lane per (landmark, value)) and the W
  // image -> global.  A FIXED number of store instructions, none under a branch (a
  // store under a branch makes the compiler's vmcnt bookkeeping wait for every load
  // in flight): lanes with nothing to store write to a dump, W passes past the end
  // repeat the last 16-byte piece (one request each).
#define LING_FLUSH_FINISH()                                                         \
  {                                                                                 \
    double sacc_ = 0.0;                                                             \
    _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) sacc_ += (u_ < no) ? vv[u_] : 0.0; \
    for (int q_ = 8; q_ < no; ++q_) sacc_ += sum_p[q_ * 9];                         \
    const int ils_ = il0p + sum_li;                                                 \
    double *dst_ = (sum_on && ils_ < nl) ? sum_g + (size_t)ils_ * sum_stride : dumpC; \
    *dst_ = sum_neg ? -sacc_ : sacc_;                                               \
    _Pragma("unroll") for (int k_ = 0; k_ < kLgPass; ++k_) dstWp[min(lane + 64 * k_, lastp)] = wvv[k_]; \
  }
#define LING_STEP(CUR, NXT2)                                                        \
  {                                                                                 \
    _Pragma("clang fp contract(fast)")                                              \
    LG_STAMP()                                                                      \
    LING_FLUSH_ISSUE()                                                              \
    {                                                                               \
      const int ilc_ = clampi(il_of(st + 2));                                       \
      uvq[NXT2] = d.obs_uv[o0 + (int64_t)ilc_ * no + oo];                           \
      const double *Xp_ = pts + (size_t)(l0 + ilc_) * 3;                            \
      Xq[NXT2][0] = LING_LDX(&Xp_[0]);                                              \
      Xq[NXT2][1] = LING_LDX(&Xp_[1]);                                              \
      Xq[NXT2][2] = LING_LDX(&Xp_[2]);                                              \
    }                                                                               \
    const int il0_ = (st * 4 + wv) * nlw;                                           \
    const int il_ = il0_ + ilw;                                                     \
    const bool valid_ = lane_on && il_ < nl && (!MASKED || uvq[CUR].x == uvq[CUR].x); \
    bool wr_ = lastw;                                                               \
    if (MASKED) {                                                                   \
      const unsigned long long vb_ = __ballot(valid_);                              \
      const unsigned mine_ = (unsigned)(vb_ >> (lane - oo));                        \
      wr_ = opt && (valid_ ? (mine_ & later_bits) == 0u : (lastw && (mine_ & pose_bits) == 0u)); \
    }                                                                               \
    ObsGeom g;                                                                      \
    project(cam, T, Xq[CUR][0], Xq[CUR][1], Xq[CUR][2], uvq[CUR].x, uvq[CUR].y, g); \
    if (MASKED) {                                                                   \
      g.r0 = valid_ ? g.r0 : 0.0;                                                   \
      g.r1 = valid_ ? g.r1 : 0.0;                                                   \
    }                                                                               \
    cost_acc += valid_ ? sqrt(g.r0 * g.r0 + g.r1 * g.r1) : 0.0;                     \
    double w, G[6], Rm[6], Q[12], cbv[9], kk[12];                                   \
    weight_and_G(cam, g, huber, w, G);                                              \
    w = valid_ ? w : 0.0;                                                           \
    make_R(G, T, Rm);                                                               \
    {                                                                               \
      /* landmark side (reference :503-517, :817-823) through w R */                \
      double wR[6];                                                                 \
      _Pragma("unroll") for (int e_ = 0; e_ < 6; ++e_) wR[e_] = w * Rm[e_];         \
      cbv[0] = Rm[0] * wR[0] + Rm[3] * wR[3];                                       \
      cbv[1] = Rm[0] * wR[1] + Rm[3] * wR[4];                                       \
      cbv[2] = Rm[0] * wR[2] + Rm[3] * wR[5];                                       \
      cbv[3] = Rm[1] * wR[1] + Rm[4] * wR[4];                                       \
      cbv[4] = Rm[1] * wR[2] + Rm[4] * wR[5];                                       \
      cbv[5] = Rm[2] * wR[2] + Rm[5] * wR[5];                                       \
      cbv[6] = wR[0] * g.r0 + wR[3] * g.r1;                                         \
      cbv[7] = wR[1] * g.r0 + wR[4] * g.r1;                                         \
      cbv[8] = wR[2] * g.r0 + wR[5] * g.r1;                                         \
      /* B_ji = w Q^T R of the pair's last-inserted observation (reference :826), */ \
      /* compact: K = w G^T R and X_ij (ba_device.h kWStride)                     */ \
      _Pragma("unroll") for (int r = 0; r < 3; ++r)                                 \
      _Pragma("unroll") for (int c = 0; c < 3; ++c)                                 \
        kk[r * 3 + c] = G[r] * wR[c] + G[3 + r] * wR[3 + c];                        \
      _Pragma("unroll") for (int r = 0; r < 3; ++r) kk[9 + r] = g.Xij[r];           \
    }                                                                               \
    {                                                                               \
      /* pose side (reference :519-556, :809); fixed poses accumulate zeros */       \
      const double wq = opt ? w : 0.0;                                              \
      make_Q(G, g.Xij, Q);                                                          \
      int k = 0;                                                                    \
      _Pragma("unroll") for (int r = 0; r < 6; ++r) {                               \
        const double q0 = wq * Q[r], q1 = wq * Q[6 + r];                            \
        _Pragma("unroll") for (int c = r; c < 6; ++c)                               \
          { acc[k] = fma(q0, Q[c], fma(q1, Q[6 + c], acc[k])); ++k; }               \
      }                                                                             \
      const double ar0 = wq * g.r0, ar1 = wq * g.r1;                                \
      _Pragma("unroll") for (int c = 0; c < 6; ++c)                                 \
        acc[21 + c] = fma(Q[c], ar0, fma(Q[6 + c], ar1, acc[21 + c]));              \
    }                                                                               \
    LG_STAMP()                                                                      \
    LING_FLUSH_FINISH()                                                             \
    LG_STAMP()                                                                      \
    /* the LDS area is free now: this step's terms take it */                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                          \
    __builtin_amdgcn_wave_barrier();                                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                          \
    {                                                                               \
      double *cb = cbw + lane * 9;                                                  \
      _Pragma("unroll") for (int e_ = 0; e_ < 9; ++e_) cb[e_] = cbv[e_];            \
      if (wr_ && lane_on) {                                                         \
        _Pragma("unroll") for (int r = 0; r < 6; ++r)                               \
          stgW_mine[r] = make_double2(kk[2 * r], kk[2 * r + 1]);                    \
      }                                                                             \
    }                                                                               \
    il0p = il0_;                                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                          \
    __builtin_amdgcn_wave_barrier();                                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                          \
  }
  int st = 0;
  while (st < nstep) {
    LING_STEP(0, 2)
    if (++st >= nstep) break;
    LING_STEP(1, 0)
    if (++st >= nstep) break;
    LING_STEP(2, 1)
    ++st;
  }
  LING_FLUSH_ISSUE()
  LING_FLUSH_FINISH()
#undef LING_STEP
#undef LING_LDX
#undef LING_FLUSH_ISSUE
#undef LING_FLUSH_FINISH
  LG_STAMP()
  {
    const double tot = block_sum(cost_acc, smc);
    if (tid == 0) d.lin_cost_part[cost_idx] = tot;
  }
  // pose-side partial sums of this piece: registers -> LDS [wave][value][lane] ->
  // one thread per (pose of the group, value); two halves of 14 values (LDS area)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 14; ++k)
      if (half * 14 + k < 27) cbw[k * 64 + lane] = acc[half * 14 + k];
    __syncthreads();
    for (int t = tid; t < dd * 14; t += kBlock) {
      const int pj = t / 14, k = t - pj * 14, e = half * 14 + k;
      if (e >= 27) continue;
      const int sb = slot_b[pj], se = slot_e[pj];
      double sacc = 0.0;
      for (int w4 = 0; w4 < 4; ++w4)
        for (int li = 0; li < nlw; ++li)
          for (int q = sb; q < se; ++q) sacc += red[w4 * kLgArea + k * 64 + li * no + q];
      d.Apart2[(size_t)(apart0 + pj) * 27 + e] = sacc;
    }
  }
  LG_STAMP()
#ifdef BA_LG_DBG
  if (lg_on) { for (int q = 0; q < 64; ++q) g_lg_dbg[q] = q < lg_n ? lg_s[q] : 0; }
#endif
}

template <bool LDSCAM, bool MASKED>
__global__ __launch_bounds__(kBlock, 2) void k_lin_grp(DevProblem d, int sel, int piece0) {
  __shared__ __attribute__((aligned(16))) double red[4 * kLgArea];
  __shared__ double cams_s[kCamLds * 16];
  __shared__ double smc[4];
  __shared__ int slot_b[kGrpMaxPoses], slot_e[kGrpMaxPoses];  // pattern slots of pose jj of the group
  lin_grp_body<LDSCAM, MASKED, false>(d, sel, piece0 + blockIdx.x, red, cams_s, smc, slot_b, slot_e);
}

// Damping and landmark inverse (reference :846-856): Cinv_i = (C_i with its
// diagonal times 1 + lambda)^-1 by the diagonally pivoted 3x3 LDL^T with
// pseudo-inverse.  One thread per landmark; runs after the control
// step has fixed lambda and the block buffer of the accepted point.
__global__ __launch_bounds__(kBlock) void k_damp_invert(DevProblem d, int store_cd) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int ic = i < d.M ? i : 0;
  const double lp1 = 1.0 + d.ctrl->lambda;
  const int lb = d.ctrl->lcur;
  const int done = d.ctrl->done;
  const double *c = d.Cu[lb] + (size_t)ic * 6;
  double cd[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) cd[k] = c[k];
  if (done || i >= d.M) return;
  cd[0] *= lp1;
  cd[3] *= lp1;
  cd[5] *= lp1;
  double ci[6];
  spd3_inverse(cd, ci);
  double *Io = d.Cinv + (size_t)i * 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) Io[k] = ci[k];
  if (store_cd) {  // readers only (ba_get_C)
    double *Do = d.Cd + (size_t)i * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) Do[k] = cd[k];
  }
}

// --------------------------------------------------------------------------
// pose side of the linearisation: one wave per chunk of one pose's
// observations -> 27 partial sums (21 upper A + 6 of Q^T w r)
// --------------------------------------------------------------------------
// Software-pipelined like k_cost: the record of observation s+128 and the
// gathered point of s+64 are in flight while s is processed.
#define LINP_STEP(XC, XN)                                                       \
  {                                                                             \
    const double *Xp_ = pts + (size_t)idn.y * 3;                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) XN[k_] = Xp_[k_];          \
    const int ncam_ = idn.x;                                                    \
    const double2 nuv_ = uvn;                                                   \
    {                                                                           \
      const int64_t s2_ = s + 128 < e ? s + 128 : e - 1;                        \
      idn = d.pobs_idx[s2_];                                                    \
      uvn = d.pobs_uv[s2_];                                                     \
    }                                                                           \
    double cam_[16];                                                            \
    load_cam<LDSCAM>(d, cams_s, ccam, cam_);                                    \
    ObsGeom g;                                                                  \
    project(cam_, Tl, XC[0], XC[1], XC[2], cuv.x, cuv.y, g);                    \
    double w, G[6], Q[12];                                                      \
    weight_and_G(cam_, g, huber, w, G);                                         \
    make_Q(G, g.Xij, Q);                                                        \
    /* reference :519-556 */                                                    \
    int k = 0;                                                                  \
    _Pragma("unroll") for (int r = 0; r < 6; ++r)                               \
    _Pragma("unroll") for (int c = r; c < 6; ++c)                               \
      { acc[k] = fma(w * Q[r], Q[c], fma(w * Q[6 + r], Q[6 + c], acc[k])); ++k; } \
    const double wr0 = w * g.r0, wr1 = w * g.r1;                                \
    _Pragma("unroll") for (int c = 0; c < 6; ++c)                               \
      acc[21 + c] = fma(Q[c], wr0, fma(Q[6 + c], wr1, acc[21 + c]));            \
    ccam = ncam_;                                                               \
    cuv = nuv_;                                                                 \
  }

template <bool LDSCAM>
__global__ __launch_bounds__(kBlock) void k_lin_poses(DevProblem d, int sel) {
  // one WAVE per chunk (observations of ONE pose, see kPoseWaveTarget): no block-level
  // synchronisation in the loop, one 6-step shuffle reduction per accumulator
  // at the end
  __shared__ double cams_s[kCamLds * 16];
  if (LDSCAM) stage_cams(d, cams_s);
  const int lane = threadIdx.x & 63;
  // wave-uniform on purpose: the chunk record and the pose (12 doubles) are then
  // fetched by scalar loads and the pose stays in SGPRs for the whole loop
  const int ch = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
  const bool live = ch < d.n_achunk;
  const int chc = live ? ch : 0;
  // chunk record and control word requested together
  const int64_t b = d.achunk_begin[chc], e = d.achunk_end[chc];
  const int j = d.achunk_pose[chc];
  const int done = d.ctrl->done;
  const int buf = sel ? d.ctrl->tcur : d.ctrl->cur;
  const double huber = d.ctrl->huber;
  BA_KEEP_S(__builtin_amdgcn_readfirstlane((int)b));
  BA_KEEP_S(__builtin_amdgcn_readfirstlane(j));
  BA_KEEP_S(done);
  __syncthreads();  // cams_s
  if (done || !live || b >= e) {
    if (!done && live) {  // empty chunk: zero partial sums
      for (int k = lane; k < 27; k += 64) d.Apart[(size_t)ch * 27 + k] = 0.0;
    }
    return;
  }
  const double *__restrict__ pts = d.pts[buf];
  const double *T = d.poses[buf] + (size_t)j * 12;
  int64_t s = b + lane;
  const int64_t s0c = s < e ? s : e - 1, s1c = s + 64 < e ? s + 64 : e - 1;
  const int2 id0 = d.pobs_idx[s0c];
  double2 cuv = d.pobs_uv[s0c];
  int2 idn = d.pobs_idx[s1c];
  double2 uvn = d.pobs_uv[s1c];
  // read through the constant address space: uniform address -> s_load, the
  // pose lives in 24 SGPRs instead of 24 VGPRs
  double Tl[12];
  {
    typedef const double __attribute__((address_space(4))) *const_f64_ptr;
    const_f64_ptr Tc = (const_f64_ptr)(uintptr_t)T;
#pragma unroll
    for (int k = 0; k < 12; ++k) Tl[k] = Tc[k];
  }
  double XA[3], XB[3];
  {
    const double *Xp = pts + (size_t)id0.y * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) XA[k] = Xp[k];
  }
  int ccam = id0.x;
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  while (s < e) {
    LINP_STEP(XA, XB)
    s += 64;
    if (s >= e) break;
    LINP_STEP(XB, XA)
    s += 64;
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const double tot = wave_sum(acc[k]);
    if (lane == 0) d.Apart[(size_t)ch * 27 + k] = tot;
  }
}

// A_j (mirrored, UNDAMPED: the readers scale the diagonal by 1 + lambda, reference
// :833-844) and a_j from the partial sums.
// One thread per (pose, component): 21 upper + 6 gradient entries.
// (the row lists are walked in batches of eight: the index loads of a batch, then its
//  value loads, are requested together — a dependent index -> value round trip per
//  row made this kernel a chain of a dozen memory latencies)
__device__ __forceinline__ void pose_finalize_body(const DevProblem &d, const int sel, const int t) {
  const int lb = sel ? d.ctrl->tlcur : d.ctrl->lcur;
  if (t >= d.N * 27) return;
  const int j = t / 27, e = t % 27;
  const int c0 = d.pose_achunk_ptr[j], c1 = d.pose_achunk_ptr[j + 1];
  const int q0 = d.pose_gpart_ptr[j], q1 = d.pose_gpart_ptr[j + 1];
  double s = 0.0;
  for (int ch = c0; ch < c1; ch += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = d.Apart[(size_t)min(ch + u, c1 - 1) * 27 + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += ch + u < c1 ? v[u] : 0.0;
  }
  // rows of the covisibility-group pieces that see this pose (k_lin_grp)
  for (int q = q0; q < q1; q += 8) {
    int row[8];
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) row[u] = d.pose_gpart[min(q + u, q1 - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = d.Apart2[(size_t)row[u] * 27 + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += q + u < q1 ? v[u] : 0.0;
  }
  if (e < 21) {
    // upper-triangle index -> (r, c)
    int r = 0, k = e;
    while (k >= 6 - r) {
      k -= 6 - r;
      ++r;
    }
    const int c = r + k;
    d.A[lb][(size_t)j * 36 + r * 6 + c] = s;
    d.A[lb][(size_t)j * 36 + c * 6 + r] = s;
  } else {
    const int r = e - 21;
    d.a[lb][(size_t)j * 6 + r] = -s;  // reference :809  a_j -= Q^T (w r)
  }
}
__global__ __launch_bounds__(kBlock) void k_pose_finalize(DevProblem d, int sel) {
  if (d.ctrl->done) return;
  pose_finalize_body(d, sel, blockIdx.x * kBlock + threadIdx.x);
}

// rhs_j = a_j - BCinv_b_j (reference :887-888), written to the packed exchange
// buffer behind the S blocks.  BCinv_b_j = sum_i V_ji b_i is accumulated next to the diagonal
// block (j,j) by the Schur kernels (entries 36..41 of every slot partial).
// One thread per (pose, component); runs after the Schur kernels.
// Landmark-major Schur complement (reference :859-872).  One workgroup per
// SUPER-RUN: a maximal run of consecutive landmarks (locality order) touching
// at most kSchurSlots distinct blocks of S.  Every block (slot) of the run is
// owned by `tps` lanes, each keeping a full 6x6 accumulator in registers for
// the whole run.  The run's W blocks stream through LDS in CHUNKS
// (<= kSchurPairs pairs): W is read from HBM exactly once, V = W Cinv is formed
// in LDS and never stored.  The loads of chunk c+1 (all contiguous, all
// independent) are issued into registers before chunk c is processed out of
// LDS, so HBM latency is hidden.  One reduction over the tps lanes and one
// 288-byte store per slot at the very end; slots of one block are summed
// across super-runs by k_schur_final in run order: deterministic, no atomics.
constexpr int kWLds = 14;  // LDS stride of a compact W record in k_schur_lds (see there)
constexpr int kSchurRW = (kSchurPairs * (kWStride / 2) + kBlock - 1) / kBlock;
constexpr int kSchurRC = (kSchurLandmarks * 3 + kBlock - 1) / kBlock;
static_assert(kSchurTri <= 4 * kBlock, "one uint4 of triple words per lane");
constexpr int kSchurRB = (kSchurLandmarks * 3 + kBlock - 1) / kBlock;

// issue the (contiguous, independent) global loads of one chunk into registers
#define SCHUR_PREFETCH(cd_)                                                   \
  {                                                                           \
    const double2 *src_ = (const double2 *)(Wg + (size_t)(cd_).p0 * kWStride);  \
    _Pragma("unroll") for (int k_ = 0; k_ < kSchurRW; ++k_) {                 \
      const int t_ = tid + k_ * kBlock;                                       \
      rw[k_] = (t_ < (cd_).np * 6) ? src_[t_] : make_double2(0.0, 0.0);       \
    }                                                                         \
    const double2 *cs_ = (const double2 *)(d.Cinv + (size_t)(cd_).l0 * 6);    \
    _Pragma("unroll") for (int k_ = 0; k_ < kSchurRC; ++k_) {                 \
      const int t_ = tid + k_ * kBlock;                                       \
      rc[k_] = (t_ < (cd_).nl * 3) ? cs_[t_] : make_double2(0.0, 0.0);        \
    }                                                                         \
    /* the chunk's triple words, four per lane (chunks start 16-byte aligned) */ \
    rt = (tid * 4 < (cd_).nt) ? ((const uint4 *)(d.ltri + (cd_).tb))[tid]     \
                              : make_uint4(0u, 0u, 0u, 0u);                   \
    _Pragma("unroll") for (int k_ = 0; k_ < kSchurRB; ++k_) {                 \
      const int t_ = tid + k_ * kBlock;                                       \
      rb[k_] = (t_ < (cd_).nl * 3) ? bg[(size_t)(cd_).l0 * 3 + t_] : 0.0;     \
    }                                                                         \
    /* raw value: any arithmetic here would wait for every load above */      \
    rpl = (tid < (cd_).np) ? d.pair_lm[(cd_).p0 + tid] : 0;                   \
    rsp = (tid <= ns) ? (int)d.chunk_sp[(cd_).sp + tid] : 0;                  \
  }

__device__ __forceinline__ int64_t uni64(int64_t v) {  // wave-uniform value -> SGPRs
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
#ifdef BA_SCHUR_DBG
// developer instrumentation (tools/schur_dbg.py): phase time stamps of four
// workgroups, kept in LDS during the run and copied out at the end
__device__ long long g_schur_dbg[4][160];
#define DBG_STAMP() { if (dbg_on && dbg_n < 160) dbg_s[dbg_n++] = clock64(); }
#else
#define DBG_STAMP()
#endif
__global__ __launch_bounds__(kBlock, 3) void k_schur_lds(DevProblem d) {
#ifdef BA_SCHUR_DBG
  __shared__ long long dbg_s[160];
  const int dbg_slot = blockIdx.x == 10 ? 0 : blockIdx.x == 700 ? 1 : blockIdx.x == 1200 ? 2 : blockIdx.x == 1900 ? 3 : -1;
  const bool dbg_on = dbg_slot >= 0 && threadIdx.x == 0;
  int dbg_n = 0;
  DBG_STAMP()
#endif
  // compact {K, X_ij} records at a stride of 14 doubles: the triple loop reads them with
  // 16-byte loads at arbitrary pair indices, and 112 bytes (28 banks, gcd with 64 = 4) spread
  // over all LDS banks where the packed 96 bytes (24 banks, gcd 8) reach only half of them
  __shared__ __attribute__((aligned(16))) double Ws[kSchurPairs * kWLds];
  __shared__ __attribute__((aligned(16))) double Vs[kSchurPairs * 18];
  __shared__ __attribute__((aligned(16))) double Cs[kSchurLandmarks * 6];
  __shared__ double Bs[kSchurLandmarks * 3];
  __shared__ __attribute__((aligned(16))) uint32_t Ts[kSchurTri];
  __shared__ uint16_t Sp[kSchurSlots + 1];
  __shared__ uint16_t Pl[kSchurPairs];
  __shared__ DevProblem::ChunkDesc Cdsc[kSchurSuperChunks];
  const int tid = threadIdx.x;
  const DevProblem::SupDesc sd = d.sup_desc[blockIdx.x];
  const uint32_t lane_word = d.sup_lane[(size_t)blockIdx.x * kBlock + tid];
  const int ns = sd.ns;
  const int done = d.ctrl->done;
  const int lbs = d.ctrl->lcur;
  BA_KEEP_S(sd.chunk_begin);
  BA_KEEP_S(done);
  if (done) return;
  const double *__restrict__ Wg = d.W[lbs];
  const double *__restrict__ bg = d.b[lbs];
  // Lane table of the run (host: deal_lanes in ba_plan.cpp): every slot owns an
  // even number of lanes inside one wave, in proportion to its triple count.
  // A lane owns HALF a slot: rows 3h..3h+2 of the 6x6 block (18 accumulators +
  // 3 of the rhs), so that the kernel fits three workgroups per CU; the tps2
  // lanes of one half share the slot's triples and are summed at the end by a
  // guarded shuffle-down tree.
  const int slot = (int)(lane_word & 0xffu);
  const int h = (int)((lane_word >> 8) & 1u), sub2 = (int)((lane_word >> 9) & 0x1fu);
  const int tps2 = (int)((lane_word >> 14) & 0x3fu);
  const bool owner = slot < ns;
  double acc[18], racc[3];
#pragma unroll
  for (int k = 0; k < 18; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) racc[k] = 0.0;
  // The run's chunk descriptors are staged in LDS once (vector loads): a scalar
  // load inside the loop would share lgkmcnt with the LDS traffic and expose a
  // full memory latency at the first LDS wait of every chunk.
  DevProblem::ChunkDesc cd = d.chunk_desc[sd.chunk_begin];
  if (tid < sd.chunk_end - sd.chunk_begin) Cdsc[tid] = d.chunk_desc[sd.chunk_begin + tid];
  double2 rw[kSchurRW], rc[kSchurRC];
  uint4 rt;
  double rb[kSchurRB];
  int rpl, rsp;
  SCHUR_PREFETCH(cd)
  __syncthreads();  // Cdsc visible
  for (int ch = sd.chunk_begin; ch < sd.chunk_end; ++ch) {
    // registers -> LDS
    {
      // compact records {K, X_ij} of the chunk's pairs, as they lie in HBM
      double2 *dst = (double2 *)Ws;
#pragma unroll
      for (int k = 0; k < kSchurRW; ++k) {
        const int t = tid + k * kBlock;
        const int pr = t / 6;
        if (t < cd.np * 6) dst[pr * (kWLds / 2) + (t - pr * 6)] = rw[k];
      }
      double2 *cdst = (double2 *)Cs;
#pragma unroll
      for (int k = 0; k < kSchurRC; ++k) {
        const int t = tid + k * kBlock;
        if (t < cd.nl * 3) cdst[t] = rc[k];
      }
      if (tid * 4 < cd.nt) ((uint4 *)Ts)[tid] = rt;
#pragma unroll
      for (int k = 0; k < kSchurRB; ++k) {
        const int t = tid + k * kBlock;
        if (t < cd.nl * 3) Bs[t] = rb[k];
      }
      if (tid < cd.np) Pl[tid] = (uint16_t)(rpl - cd.l0);
      if (tid <= ns) Sp[tid] = (uint16_t)rsp;
    }
    const int np = cd.np;
    DBG_STAMP()
    if (ch + 1 < sd.chunk_end) {  // next chunk's loads fly during this one
      const DevProblem::ChunkDesc *q = &Cdsc[ch + 1 - sd.chunk_begin];
      cd.p0 = uni64(q->p0);
      cd.tb = uni64(q->tb);
      cd.sp = uni64(q->sp);
      cd.l0 = __builtin_amdgcn_readfirstlane(q->l0);
      cd.nl = __builtin_amdgcn_readfirstlane(q->nl);
      cd.np = __builtin_amdgcn_readfirstlane(q->np);
      cd.nt = __builtin_amdgcn_readfirstlane(q->nt);
      SCHUR_PREFETCH(cd)
    }
    DBG_STAMP()
    __syncthreads();
    DBG_STAMP()
    // V = W Cinv from LDS: thread p forms rows 0..2 (W rows 0..2 = K), thread
    // 128 + p rows 3..5 (W rows 3..5 = X_ij x columns of K, formed in registers:
    // the triple loop below rebuilds them from {K, X_ij} as well, so they are
    // never stored).  kBlock == 2 * kSchurPairs.
    {
      static_assert(kBlock == 2 * kSchurPairs && (kSchurPairs & (kSchurPairs - 1)) == 0, "V phase mapping");
      const int pr = tid & (kSchurPairs - 1);
      const bool hi = tid >= kSchurPairs;  // wave-uniform
      if (pr < np) {
        const double *kk = Ws + pr * kWLds;
        const double *ci = Cs + (int)Pl[pr] * 6;
        double k9[9], c6[6], x3[3], wr[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) k9[e] = kk[e];
#pragma unroll
        for (int e = 0; e < 6; ++e) c6[e] = ci[e];
        if (hi) {
#pragma unroll
          for (int e = 0; e < 3; ++e) x3[e] = kk[9 + e];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            wr[c] = x3[1] * k9[6 + c] - x3[2] * k9[3 + c];
            wr[3 + c] = x3[2] * k9[c] - x3[0] * k9[6 + c];
            wr[6 + c] = x3[0] * k9[3 + c] - x3[1] * k9[c];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 9; ++e) wr[e] = k9[e];
        }
        double *vo = Vs + pr * 18 + (hi ? 9 : 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double w0 = wr[r * 3], w1 = wr[r * 3 + 1], w2 = wr[r * 3 + 2];
          vo[r * 3 + 0] = w0 * c6[0] + w1 * c6[1] + w2 * c6[2];
          vo[r * 3 + 1] = w0 * c6[1] + w1 * c6[3] + w2 * c6[4];
          vo[r * 3 + 2] = w0 * c6[2] + w1 * c6[4] + w2 * c6[5];
        }
      }
    }
    DBG_STAMP()
    __syncthreads();
    DBG_STAMP()
    if (owner) {
      // The triple word carries everything the iteration needs (pair p, pair q,
      // local landmark), and the word of the NEXT iteration is read one
      // iteration ahead: otherwise every iteration starts with two dependent
      // LDS round trips (word -> landmark index -> b) before its 27 reads.
      const int t1 = (int)Sp[slot + 1];
      int t = (int)Sp[slot] + sub2;
      uint32_t pq = Ts[t < kSchurTri ? t : kSchurTri - 1];
      while (t < t1) {
        const int tn = t + tps2;
        const uint32_t pqn = Ts[tn < kSchurTri ? tn : kSchurTri - 1];  // unused past t1
        __builtin_amdgcn_sched_barrier(0);  // keep the read up here
        const uint32_t pp = pq >> 16, qq = pq & 0xffu;
        const double *vp = Vs + pp * 18 + h * 9;  // rows 3h..3h+2 of V
        // W_q from its compact record: rows 0..2 are K, rows 3..5 are X x K[:,m], so
        //   sum_m v_m W[c][m]      = t_c           (c < 3),  t = K v
        //   sum_m v_m W[3 + a][m]  = (X x t)_a
        // 12 instead of 18 LDS doubles per triple; the loop is LDS-bandwidth bound
        const double2 *wp = (const double2 *)(Ws + qq * kWLds);
        // diagonal triple (p == q): also B Cinv b of this pair (reference :864);
        // b is read unconditionally and masked (no divergent LDS reads)
        const double *bp = Bs + ((pq >> 8) & 0xffu) * 3;
        const double bm = (pp == qq) ? 1.0 : 0.0;
        const double b0 = bp[0] * bm, b1 = bp[1] * bm, b2 = bp[2] * bm;
        double w[12];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double2 b2 = wp[k];
          w[2 * k] = b2.x;
          w[2 * k + 1] = b2.y;
        }
        const double X0 = w[9], X1 = w[10], X2 = w[11];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double v0 = vp[r * 3 + 0], v1 = vp[r * 3 + 1], v2 = vp[r * 3 + 2];
          const double t0 = fma(v2, w[2], fma(v1, w[1], v0 * w[0]));
          const double t1 = fma(v2, w[5], fma(v1, w[4], v0 * w[3]));
          const double t2 = fma(v2, w[8], fma(v1, w[7], v0 * w[6]));
          acc[r * 6 + 0] += t0;
          acc[r * 6 + 1] += t1;
          acc[r * 6 + 2] += t2;
          acc[r * 6 + 3] = fma(-X2, t1, fma(X1, t2, acc[r * 6 + 3]));
          acc[r * 6 + 4] = fma(-X0, t2, fma(X2, t0, acc[r * 6 + 4]));
          acc[r * 6 + 5] = fma(-X1, t0, fma(X0, t1, acc[r * 6 + 5]));
          racc[r] = fma(v2, b2, fma(v1, b1, fma(v0, b0, racc[r])));
        }
        pq = pqn;
        t = tn;
      }
    }
    DBG_STAMP()
    __syncthreads();  // LDS is rewritten by the next chunk
    DBG_STAMP()
  }
  DBG_STAMP()
  // sum the tps/2 lanes of every half slot: guarded shuffle-down tree over the
  // lanes of equal h (stride 2; fixed order, any tps/2 <= 16), result in the
  // half's first lane
#pragma unroll
  for (int k = 0; k < 21; ++k) {
    double a2 = k < 18 ? acc[k] : racc[k - 18];
    int w = tps2;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      const double o2 = __shfl_down(a2, 2 * off, 64);
      if (sub2 + off < w) a2 += o2;
      w = w < off ? w : off;
    }
    if (k < 18) acc[k] = a2; else racc[k - 18] = a2;
  }
#ifdef BA_SCHUR_DBG
  if (dbg_on)
    for (int k = 0; k < 160; ++k) g_schur_dbg[dbg_slot][k] = k < dbg_n ? dbg_s[k] : 0;
#endif
  if (owner && sub2 == 0) {  // 144 + 24 B per half slot
    double *o = d.spart2 + (size_t)(sd.s0 + slot) * kSlotStride;
#pragma unroll
    for (int k = 0; k < 18; ++k) o[h * 18 + k] = acc[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[36 + h * 3 + k] = racc[k];
  }
}

// --------------------------------------------------------------------------
// Covisibility groups (host: ba_plan.cpp): n landmarks seen by the IDENTICAL set
// of d optimisable poses.  Their part of the Schur complement (reference
// :859-872) is, for every pose pair (jj, kk) of the set,
//     sum_i V_(jj)i W_(kk)i^T ,   V = W Cinv,
// i.e. ONE dense product  [V]^T [W]  with  [V], [W] : (3 n) x (6 d)  (row 3 i + m
// = landmark i, coordinate m; column 6 jj + r = pose jj, row r of its 6x3 block),
// and the right-hand side part sum_i V_ji b_i is one more column (b) of [W].
// The product runs on v_mfma_f64_16x16x4_f64: the super-run kernel's triple loop
// is bound by LDS bandwidth (24 doubles per 63 FMAs and lane); here an LDS double
// feeds 16 FMAs.
// One workgroup per group (or piece of a large group); its four waves work
// INDEPENDENTLY on every fourth chunk of floor(64 / d) landmarks (one lane per
// (landmark, pose) pair): the lane turns its compact W record and Cinv_i into its
// 6x3 blocks of [V] and [W] in the wave's private LDS image, then the wave
// multiplies the image into its NT (NT + 1) / 2 upper 16x16 accumulator tiles.  No
// workgroup barrier inside the loop (a wave's LDS operations are ordered), so the
// VALU / LDS-store phase of one wave overlaps the MFMA phase of the others; the
// operands of the next two chunks are in flight meanwhile (register ring).  The
// four partial tiles are added in wave order at the very end (deterministic) and
// leave as d (d + 1) / 2 slot partials that k_schur_final sums with the super-run
// slots.
// LDS image: entry (k, col) at (k >> 1) * RS + col * 2 + (k & 1): the four k rows
// of one MFMA operand read (lanes: col = lane & 15, k = 4 ks + (lane >> 4)) are two
// 256-byte rows of 16-byte (k even, k odd) cells — two conflict-free passes; the
// padding between row pairs (RS) is for the STORES that build the image.
#ifndef BA_GRP_KRW
#define BA_GRP_KRW 36
#endif
#ifdef BA_GRP_DBG
__device__ long long g_grp_dbg[256];
#define GRP_STAMP() { if (gdbg_on && gdbg_n < 256) gdbg_s[gdbg_n++] = clock64(); }
#else
#define GRP_STAMP()
#endif
template <int NT>
__global__ __launch_bounds__(kBlock, (NT == 2 && BA_GRP_KRW <= 24) ? 3 : 2) void k_schur_grp(DevProblem d, const DevProblem::GrpDesc *grps) {
  constexpr int TW = 16 * NT;             // padded width: 6 d (+ 1 for b) <= TW
#ifndef BA_GRP_KRW
#define BA_GRP_KRW 36
#endif
  constexpr int KRW = NT == 2 ? BA_GRP_KRW : 20;  // k rows of a wave's image (multiple of 4)
  constexpr int NTILE = NT * (NT + 1) / 2;
  // doubles per pair of k rows: 2 TW cells + padding.  Without it the row pairs are 128
  // dwords apart, the bank of a lane's store depends on its pose only and the
  // landmarks of a chunk collide (6-way); 6 doubles spread them over the banks (2
  // lanes per bank: the minimum for 64 x 8 bytes).  (NT = 4: no room for padding at
  // two workgroups per CU.)
  constexpr int RS = 2 * TW + (NT == 2 ? 6 : 0);
  constexpr int IMG = (KRW / 2) * RS;
  __shared__ __attribute__((aligned(16))) double VA[4][IMG];
  __shared__ __attribute__((aligned(16))) double WB[4][IMG];
  static_assert(IMG >= 256, "the final reduction uses VA as 4 x 256 doubles");
#ifdef BA_GRP_DBG
  __shared__ long long gdbg_s[256];
  const bool gdbg_on = blockIdx.x == 600 && threadIdx.x == 0;
  int gdbg_n = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const DevProblem::GrpDesc gd = grps[blockIdx.x];
  const int done = d.ctrl->done;
  const int lb = d.ctrl->lcur;
  BA_KEEP_S(gd.l0);
  BA_KEEP_S(done);
  if (done) return;
  const double *__restrict__ Wg = d.W[lb];
  const double *__restrict__ bg = d.b[lb];
  const double *__restrict__ Cug = d.Cu[lb];
  const double lp1 = 1.0 + d.ctrl->lambda;
  const int dd = gd.d;
  const int nlw = min(64 / dd, KRW / 3);       // landmarks per wave chunk
  const int il = lane / dd, jj = lane - il * dd;  // this lane's pair inside a chunk
  const int nch = (gd.nl + nlw - 1) / nlw;
  double *va = VA[wv], *wb = WB[wv];
  for (int e = lane; e < IMG; e += 64) {  // rows a full chunk never writes stay zero
    va[e] = 0.0;
    wb[e] = 0.0;
  }
  // operands of a chunk, requested two chunks ahead (register ring; the loop is
  // unrolled by two so that the ring index is static)
  double2 rw[2][6], rc[2][3];
  double rb[2][3];
#define GRP_PREFETCH(B, ch_)                                                        \
  {                                                                                 \
    const int c0_ = (ch_) * nlw;                                                    \
    const bool on_ = il < min(nlw, gd.nl - c0_);                                    \
    const double2 *wp_ = (const double2 *)(Wg + (size_t)(gd.p0 + (int64_t)c0_ * dd + (on_ ? lane : 0)) * kWStride); \
    const int lm_ = gd.l0 + c0_ + (on_ ? il : 0);                                   \
    const double2 *cp_ = (const double2 *)(Cug + (size_t)lm_ * 6);                  \
    _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) rw[B][k_] = wp_[k_];           \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) rc[B][k_] = cp_[k_];           \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) rb[B][k_] = bg[(size_t)lm_ * 3 + k_]; \
  }
  v4f64 acc[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (wv < nch) GRP_PREFETCH(0, wv)
  if (wv + 4 < nch) GRP_PREFETCH(1, wv + 4)
#define GRP_IDX(k_, col_) (((k_) >> 1) * RS + (col_) * 2 + ((k_) & 1))
#define GRP_STAGE(B)                                                                \
  if (ch < nch) {                                                                   \
    GRP_STAMP()                                                                     \
    const int nlc = min(nlw, gd.nl - ch * nlw);                                     \
    if (il < nlc) {                                                                 \
      /* W_p (6x3) from its compact record, V_p = W_p Cinv_i (fused multiply-adds: the fp64 \
         VALU shares its pipe with the fp64 MFMA on this part, every instruction counts) */ \
      _Pragma("clang fp contract(fast)")                                            \
      const double k9[9] = {rw[B][0].x, rw[B][0].y, rw[B][1].x, rw[B][1].y, rw[B][2].x, \
                            rw[B][2].y, rw[B][3].x, rw[B][3].y, rw[B][4].x};        \
      const double x0 = rw[B][4].y, x1 = rw[B][5].x, x2 = rw[B][5].y;               \
      /* damping and the 3x3 inverse (reference :846-856) in the lane: the d lanes of a \
         landmark repeat it, which is cheaper than a kernel + an array for Cinv */    \
      const double cd_[6] = {rc[B][0].x * lp1, rc[B][0].y, rc[B][1].x, rc[B][1].y * lp1, rc[B][2].x, rc[B][2].y * lp1}; \
      double c6[6];                                                                 \
      spd3_inverse(cd_, c6);                                                        \
      double w[18];                                                                 \
      _Pragma("unroll") for (int e = 0; e < 9; ++e) w[e] = k9[e];                   \
      _Pragma("unroll") for (int c = 0; c < 3; ++c) {                               \
        w[9 + c] = x1 * k9[6 + c] - x2 * k9[3 + c];                                 \
        w[12 + c] = x2 * k9[c] - x0 * k9[6 + c];                                    \
        w[15 + c] = x0 * k9[3 + c] - x1 * k9[c];                                    \
      }                                                                             \
      const int kb = 3 * il, cb = 6 * jj;                                           \
      const int i0 = GRP_IDX(kb, cb), i1 = GRP_IDX(kb + 1, cb), i2 = GRP_IDX(kb + 2, cb); \
      _Pragma("unroll") for (int r = 0; r < 6; ++r) {                               \
        const double w0 = w[r * 3], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];           \
        va[i0 + 2 * r] = w0 * c6[0] + w1 * c6[1] + w2 * c6[2];                      \
        va[i1 + 2 * r] = w0 * c6[1] + w1 * c6[3] + w2 * c6[4];                      \
        va[i2 + 2 * r] = w0 * c6[2] + w1 * c6[4] + w2 * c6[5];                      \
        wb[i0 + 2 * r] = w0;                                                        \
        wb[i1 + 2 * r] = w1;                                                        \
        wb[i2 + 2 * r] = w2;                                                        \
      }                                                                             \
      if (jj == 0) { /* the b column of [W]: column 6 d */                          \
        wb[GRP_IDX(kb, 6 * dd)] = rb[B][0];                                         \
        wb[GRP_IDX(kb + 1, 6 * dd)] = rb[B][1];                                     \
        wb[GRP_IDX(kb + 2, 6 * dd)] = rb[B][2];                                     \
      }                                                                             \
    }                                                                               \
    const int nks = (3 * nlc + 3) >> 2;                                             \
    if (nlc < nlw) { /* partial last chunk: stale rows up to the next multiple of four */ \
      for (int e = lane; e < (4 * nks - 3 * nlc) * TW; e += 64) {                   \
        const int k_ = 3 * nlc + e / TW, c_ = e - (e / TW) * TW;                    \
        va[GRP_IDX(k_, c_)] = 0.0;                                                  \
        wb[GRP_IDX(k_, c_)] = 0.0;                                                  \
      }                                                                             \
    }                                                                               \
    GRP_STAMP()                                                                     \
    if (ch + 8 < nch) GRP_PREFETCH(B, ch + 8)                                       \
    GRP_STAMP()                                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                          \
    __builtin_amdgcn_wave_barrier();                                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                          \
    {                                                                               \
      /* operands of k step ks: rows 4 ks + lk -> cell row 2 ks + (lk >> 1), half lk & 1 */ \
      const double *ap = va + (lk >> 1) * RS + lr * 2 + (lk & 1);                   \
      const double *bp = wb + (lk >> 1) * RS + lr * 2 + (lk & 1);                   \
      double a[NT], b[NT], an[NT], bn[NT];                                          \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) {                              \
        a[t] = ap[32 * t];                                                          \
        b[t] = bp[32 * t];                                                          \
      }                                                                             \
      for (int ks = 0; ks < nks; ++ks) {                                            \
        const int kn = ks + 1 < nks ? ks + 1 : ks; /* operands of the next step first */ \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) {                            \
          an[t] = ap[kn * 2 * RS + 32 * t];                                         \
          bn[t] = bp[kn * 2 * RS + 32 * t];                                         \
        }                                                                           \
        int tile = 0;                                                               \
        _Pragma("unroll") for (int ti = 0; ti < NT; ++ti)                           \
        _Pragma("unroll") for (int tj = ti; tj < NT; ++tj) {                        \
          acc[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[tile], 0, 0, 0); \
          ++tile;                                                                   \
        }                                                                           \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) {                            \
          a[t] = an[t];                                                             \
          b[t] = bn[t];                                                             \
        }                                                                           \
      }                                                                             \
    }                                                                               \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                          \
    __builtin_amdgcn_wave_barrier();                                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                          \
    GRP_STAMP()                                                                     \
    ch += 4;                                                                        \
  }
  for (int ch = wv; ch < nch;) {
    GRP_STAGE(0)
    GRP_STAGE(1)
  }
#undef GRP_STAGE
#undef GRP_IDX
#undef GRP_PREFETCH
  GRP_STAMP()
#ifdef BA_GRP_DBG
  if (gdbg_on) for (int q = 0; q < 256; ++q) g_grp_dbg[q] = q < gdbg_n ? gdbg_s[q] : 0;
#endif
  // add the four waves' partial tiles in wave order, one tile at a time, and
  // scatter the block entries to the group's slots
  __syncthreads();
  double *R = &VA[0][0];  // 4 x 256 doubles
  int tile = 0;
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int tj = ti; tj < NT; ++tj) {
#pragma unroll
      for (int g = 0; g < 4; ++g) R[wv * 256 + g * 64 + lane] = acc[tile][g];
      __syncthreads();
      {
        const double v = ((R[tid] + R[256 + tid]) + R[512 + tid]) + R[768 + tid];
        const int g = tid >> 6, ln = tid & 63;
        const int row = 16 * ti + (ln >> 4) + 4 * g;  // V side: 6 pj + r
        const int col = 16 * tj + (ln & 15);          // W side: 6 pk + c, or the b column
        const int pj = row / 6, r = row - 6 * pj;
        if (pj < dd) {
          const int sj = gd.s0 + pj * dd - (pj * (pj - 1)) / 2;  // slot of (pj, pj)
          if (col == 6 * dd) {
            d.spart2[(size_t)sj * kSlotStride + 36 + r] = v;
          } else if (col < 6 * dd) {
            const int pk = col / 6, c = col - 6 * pk;
            if (pk >= pj) d.spart2[(size_t)(sj + (pk - pj)) * kSlotStride + r * 6 + c] = v;
          }
        }
      }
      __syncthreads();
      ++tile;
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the
// global loads in flight (s_waitcnt vmcnt(0)): with the operands of the next two stages
// requested just before it, every barrier would cost one memory latency (measured: 4.8-7 k
// cycles per stage of k_schur_grp_wide).
#define GW_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// Pose sets of 11..20 poses (6 d + 1 <= 121 columns: a 128-wide image, 36 upper 16x16
// tiles).  144 accumulator registers per wave do not fit beside the staging arithmetic,
// so here the four waves split the TILES, not the chunks: a stage = 4 x nlw landmarks
// (nlw = min(64 / d, 2) per wave, one lane per (landmark, pose) pair) staged by all waves
// into ONE shared image of <= 24 k rows; after a workgroup barrier wave w multiplies
// the image into ITS nine tiles — tile rows w and 7 - w of the upper triangle — so
// every tile has one owner, the operands of a k step are 2 + 9 LDS reads per lane for 9
// MFMAs, and there is no final sum over waves: a wave scatters its tiles to the group's
// slots itself.  Same arithmetic as k_schur_grp per entry; the k order is the landmark
// order (deterministic).
__global__ __launch_bounds__(kBlock, 2) void k_schur_grp_wide(DevProblem d, const DevProblem::GrpDesc *grps) {
  constexpr int TW = 128;
  constexpr int LPW = 2;            // landmarks a wave stages per stage
  constexpr int KRW = 4 * LPW * 3;  // k rows of a stage
  constexpr int RS = 2 * TW + 6;    // doubles per pair of k rows (+ padding: see k_schur_grp)
  constexpr int IMG = (KRW / 2) * RS;
  __shared__ __attribute__((aligned(16))) double VA[IMG];
  __shared__ __attribute__((aligned(16))) double WB[IMG];
#ifdef BA_GRP_DBG
  __shared__ long long gdbg_s[256];
  #ifndef BA_GRP_DBG_TID
#define BA_GRP_DBG_TID 0
#endif
  const bool gdbg_on = blockIdx.x == 100 && threadIdx.x == BA_GRP_DBG_TID;
  int gdbg_n = 0;
#endif
  GRP_STAMP()
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const DevProblem::GrpDesc *gp = grps + blockIdx.x;
  const int64_t gp0 = gp->p0;
  const int gl0 = gp->l0, gnl = gp->nl, dd = gp->d, gs0 = gp->s0;
  const int done = d.ctrl->done;
  const int lb = d.ctrl->lcur;
  if (done) return;
  const double *__restrict__ Wg = d.W[lb];
  const double *__restrict__ bg = d.b[lb];
  const double *__restrict__ Cug = d.Cu[lb];
  const double lp1 = 1.0 + d.ctrl->lambda;
  const int nlw = min(64 / dd, LPW);
  const int il = lane / dd, jj = lane - il * dd;
  const int li = wv * nlw + il;  // this lane's landmark inside a stage
  const int per_stage = 4 * nlw;
  const int nst = (gnl + per_stage - 1) / per_stage;
  for (int e = tid; e < IMG; e += kBlock) {  // rows a full stage never writes stay zero
    VA[e] = 0.0;
    WB[e] = 0.0;
  }
  // operands of the NEXT stage, requested while this one is multiplied (one buffer: a two-
  // deep ring made the compiler copy in-flight registers across the loop and wait for them)
  double2 rw[1][6], rc[1][3];
  double rb[1][3];
#define GW_PREFETCH(B, st_)                                                         \
  {                                                                                 \
    const int c0_ = (st_) * per_stage;                                              \
    const bool on_ = il < nlw && c0_ + li < gnl;                                     \
    const double2 *wp_ = (const double2 *)(Wg + (size_t)(gp0 + (int64_t)(c0_ + wv * nlw) * dd + (on_ ? lane : 0)) * kWStride); \
    const int lm_ = gl0 + (on_ ? c0_ + li : 0);                                     \
    const double2 *cp_ = (const double2 *)(Cug + (size_t)lm_ * 6);                  \
    if (!on_) wp_ = (const double2 *)(Wg + (size_t)gp0 * kWStride);                 \
    _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) rw[B][k_] = wp_[k_];           \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) rc[B][k_] = cp_[k_];           \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) rb[B][k_] = bg[(size_t)lm_ * 3 + k_]; \
  }
  v4f64 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
  // this wave's tiles: t < n_first: (ti, tj) = (wv, wv + t); else (7 - wv, 7 - wv + t - n_first)
  const int n_first = 8 - wv;
  GW_PREFETCH(0, 0)
#define GW_IDX(k_, col_) (((k_) >> 1) * RS + (col_) * 2 + ((k_) & 1))
#define GW_STAGE(B)                                                                 \
  {                                                                                 \
    const int c0 = st * per_stage;                                                  \
    const int nlc = min(per_stage, gnl - c0);                                       \
    /* the stage's operands are waited for HERE, on every path: consumed only under \
       the branch below, they would count as still in flight behind it and the next \
       prefetch (which reuses their registers) would wait for ALL loads */           \
    _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) asm volatile("" : "+v"(rw[B][k_].x), "+v"(rw[B][k_].y)); \
    _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) asm volatile("" : "+v"(rc[B][k_].x), "+v"(rc[B][k_].y), "+v"(rb[B][k_])); \
    if (il < nlw && li < nlc) {                                                     \
      _Pragma("clang fp contract(fast)")                                            \
      const double k9[9] = {rw[B][0].x, rw[B][0].y, rw[B][1].x, rw[B][1].y, rw[B][2].x, \
                            rw[B][2].y, rw[B][3].x, rw[B][3].y, rw[B][4].x};        \
      const double x0 = rw[B][4].y, x1 = rw[B][5].x, x2 = rw[B][5].y;               \
      const double cd_[6] = {rc[B][0].x * lp1, rc[B][0].y, rc[B][1].x, rc[B][1].y * lp1, rc[B][2].x, rc[B][2].y * lp1}; \
      double c6[6];                                                                 \
      spd3_inverse(cd_, c6);                                                        \
      double w[18];                                                                 \
      _Pragma("unroll") for (int e = 0; e < 9; ++e) w[e] = k9[e];                   \
      _Pragma("unroll") for (int c = 0; c < 3; ++c) {                               \
        w[9 + c] = x1 * k9[6 + c] - x2 * k9[3 + c];                                 \
        w[12 + c] = x2 * k9[c] - x0 * k9[6 + c];                                    \
        w[15 + c] = x0 * k9[3 + c] - x1 * k9[c];                                    \
      }                                                                             \
      const int kb = 3 * li, cb = 6 * jj;                                           \
      const int i0 = GW_IDX(kb, cb), i1 = GW_IDX(kb + 1, cb), i2 = GW_IDX(kb + 2, cb); \
      _Pragma("unroll") for (int r = 0; r < 6; ++r) {                               \
        const double w0 = w[r * 3], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];           \
        VA[i0 + 2 * r] = w0 * c6[0] + w1 * c6[1] + w2 * c6[2];                      \
        VA[i1 + 2 * r] = w0 * c6[1] + w1 * c6[3] + w2 * c6[4];                      \
        VA[i2 + 2 * r] = w0 * c6[2] + w1 * c6[4] + w2 * c6[5];                      \
        WB[i0 + 2 * r] = w0;                                                        \
        WB[i1 + 2 * r] = w1;                                                        \
        WB[i2 + 2 * r] = w2;                                                        \
      }                                                                             \
      if (jj == 0) {                                                                \
        WB[GW_IDX(kb, 6 * dd)] = rb[B][0];                                          \
        WB[GW_IDX(kb + 1, 6 * dd)] = rb[B][1];                                      \
        WB[GW_IDX(kb + 2, 6 * dd)] = rb[B][2];                                      \
      }                                                                             \
    }                                                                               \
    GRP_STAMP()                                                                     \
    const int nks = (3 * nlc + 3) >> 2;                                             \
    if (nlc < per_stage) { /* partial last stage: stale rows up to the next multiple of four */ \
      for (int e = tid; e < (4 * nks - 3 * nlc) * TW; e += kBlock) {                \
        const int k_ = 3 * nlc + e / TW, c_ = e - (e / TW) * TW;                    \
        VA[GW_IDX(k_, c_)] = 0.0;                                                   \
        WB[GW_IDX(k_, c_)] = 0.0;                                                   \
      }                                                                             \
    }                                                                               \
    /* (unconditional, clamped: behind a branch the compiler's load counters merge the \
        path without the prefetch and every stage waits for ALL loads in flight) */    \
    GW_PREFETCH(B, min(st + 1, nst - 1))                                            \
    GRP_STAMP()                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
    GRP_STAMP()                                                                     \
    GW_LDS_BARRIER();                                                                \
    GRP_STAMP()                                                                     \
    {                                                                               \
      const double *ap = VA + (lk >> 1) * RS + lr * 2 + (lk & 1);                   \
      const double *bp = WB + (lk >> 1) * RS + lr * 2 + (lk & 1);                   \
      for (int ks = 0; ks < nks; ++ks) {                                            \
        const double a0 = ap[ks * 2 * RS + 32 * wv], a1 = ap[ks * 2 * RS + 32 * (7 - wv)]; \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) {                             \
          const bool first = t < n_first;                                           \
          const int tj = first ? wv + t : 7 - wv + (t - n_first);                   \
          const double bv = bp[ks * 2 * RS + 32 * tj];                              \
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(first ? a0 : a1, bv, acc[t], 0, 0, 0); \
        }                                                                           \
      }                                                                             \
    }                                                                               \
    GRP_STAMP()                                                                     \
    GW_LDS_BARRIER();                                                                \
    GRP_STAMP()                                                                     \
    ++st;                                                                           \
  }
  GW_LDS_BARRIER();
  for (int st = 0; st < nst;) {
    GW_STAGE(0)
  }
#undef GW_STAGE
#undef GW_IDX
#undef GW_PREFETCH
  GRP_STAMP()
  // every tile has one owner: the wave scatters its nine tiles to the group's slots
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const bool first = t < n_first;
    const int ti = first ? wv : 7 - wv;
    const int tj = first ? wv + t : 7 - wv + (t - n_first);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const double v = acc[t][g];
      const int row = 16 * ti + lk + 4 * g;  // V side: 6 pj + r
      const int col = 16 * tj + lr;          // W side: 6 pk + c, or the b column
      const int pj = row / 6, r = row - 6 * pj;
      if (pj < dd) {
        const int sj = gs0 + pj * dd - (pj * (pj - 1)) / 2;  // slot of (pj, pj)
        if (col == 6 * dd) {
          d.spart2[(size_t)sj * kSlotStride + 36 + r] = v;
        } else if (col < 6 * dd) {
          const int pk = col / 6, c = col - 6 * pk;
          if (pk >= pj) d.spart2[(size_t)(sj + (pk - pj)) * kSlotStride + r * 6 + c] = v;
        }
      }
    }
  }
  GRP_STAMP()
#ifdef BA_GRP_DBG
  if (gdbg_on) for (int q = 0; q < 256; ++q) g_grp_dbg[q] = q < gdbg_n ? gdbg_s[q] : 0;
#endif
}

// Same sums for landmarks seen by more than kSchurPairs poses: one wave per
// chunk of the block's global triple list, V computed on the fly.
__global__ __launch_bounds__(64) void k_schur_partial(DevProblem d) {
  if (d.ctrl->done) return;
  const double *__restrict__ Wg = d.W[d.ctrl->lcur];
  const double *__restrict__ bg = d.b[d.ctrl->lcur];
  const int ch = blockIdx.x;
  double acc[36], racc[6];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) racc[k] = 0.0;
  const int64_t e = d.tchunk_end[ch];
  for (int64_t t = d.tchunk_begin[ch] + threadIdx.x; t < e; t += 64) {
    const int64_t p = d.tri_p[t];
    const bool dg = p == d.tri_q[t];
    double Wp[18];
    expand_W(Wg + (size_t)p * kWStride, Wp);
    const double *ci = d.Cinv + (size_t)d.pair_lm[p] * 6;
    const double *bi = bg + (size_t)d.pair_lm[p] * 3;
    const double b0 = dg ? bi[0] : 0.0, b1 = dg ? bi[1] : 0.0, b2 = dg ? bi[2] : 0.0;
    double v[18], w[18];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double w0 = Wp[r * 3 + 0], w1 = Wp[r * 3 + 1], w2 = Wp[r * 3 + 2];
      v[r * 3 + 0] = w0 * ci[0] + w1 * ci[1] + w2 * ci[2];
      v[r * 3 + 1] = w0 * ci[1] + w1 * ci[3] + w2 * ci[4];
      v[r * 3 + 2] = w0 * ci[2] + w1 * ci[4] + w2 * ci[5];
      racc[r] += v[r * 3 + 0] * b0 + v[r * 3 + 1] * b1 + v[r * 3 + 2] * b2;
    }
    expand_W(Wg + (size_t)d.tri_q[t] * kWStride, w);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)
        acc[r * 6 + c] += v[r * 3 + 0] * w[c * 3 + 0] +
                          v[r * 3 + 1] * w[c * 3 + 1] +
                          v[r * 3 + 2] * w[c * 3 + 2];
  }
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    const double tot = wave_sum(acc[k]);
    if (threadIdx.x == 0) d.spart[(size_t)ch * kSlotStride + k] = tot;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double tot = wave_sum(racc[k]);
    if (threadIdx.x == 0) d.spart[(size_t)ch * kSlotStride + 36 + k] = tot;
  }
}

// S_jk = delta_jk A_j - BCinvBt_jk into the packed exchange buffer (reference
// :878-888).  One workgroup per block: 7 parts x 36
// entries; part q sums every 7th slot partial of the block, the parts are then
// added in order 0..6 (fixed tree: deterministic).
// One workgroup per block: 6 parts x 42 entries (36 of S; for a diagonal block
// also the 6 of B Cinv b that ride behind them in every slot partial).  Part q
// sums every 6th slot partial of the block, the parts are then added in order
// 0..5 (fixed tree: deterministic).  rhs_j = a_j - B Cinv b (reference :887-888)
// goes behind the S blocks of the packed buffer.
template <bool DIRECT>
__device__ __forceinline__ void schur_final_body(const DevProblem &d, int64_t blk,
                                                 double (*part)[42]) {
  const int q = threadIdx.x / 42, e = threadIdx.x - q * 42;
  // ONE record per block: poses, contribution range and the first eight slots
  const int4 *bq = (const int4 *)(d.blk_desc + 16 * (size_t)blk);
  const int4 b0 = bq[0], b1 = bq[1], b2 = bq[2], b3 = bq[3];
  const int done = d.ctrl->done;
  if (done) return;
  const int j = b0.x, k = b0.y;
  // what the 42 finishing threads need besides the sums depends on (j, k) only:
  // requested now, one load level earlier than after the barrier
  double pre = 0.0;
  int pcj = 0, pck = 0;
  if (threadIdx.x < 42 && (e < 36 || j == k)) {
    const int lb = d.ctrl->lcur;
    if (e >= 36) pre = d.a[lb][(size_t)j * 6 + (e - 36)];
    else if (j == k) {
      pre = d.A[lb][(size_t)j * 36 + e];
      if (e % 7 == 0) pre *= (1.0 + d.ctrl->lambda);  // damped diagonal, reference :833-844
    }
    pcj = d.pose_col[j];
    pck = d.pose_col[k];
  }
  if (q < 6 && (e < 36 || j == k)) {
    double s = 0.0;
    const int nc = b0.w;
    // contributions q, q + 6, ...: the first eight slots are in the record
    if (q < nc) {
      const int sl = q == 0 ? b2.x : q == 1 ? b2.y : q == 2 ? b2.z : q == 3 ? b2.w : q == 4 ? b3.x : b3.y;
      s += d.spart2[(size_t)sl * kSlotStride + e];
    }
    if (q + 6 < nc) {
      const int sl = q == 0 ? b3.z : q == 1 ? b3.w : d.contrib_slot[(size_t)b0.z + q + 6];
      s += d.spart2[(size_t)sl * kSlotStride + e];
    }
    for (int ci = q + 12; ci < nc; ci += 6)
      s += d.spart2[(size_t)d.contrib_slot[(size_t)b0.z + ci] * kSlotStride + e];
    const int ch1 = b1.x + b1.y;
    for (int ch = b1.x + q; ch < ch1; ch += 6)
      s += d.spart[(size_t)ch * kSlotStride + e];
    part[q][e] = s;
  }
  __syncthreads();
  if (threadIdx.x >= 42 || (e >= 36 && j != k)) return;
  double s = part[0][e];
#pragma unroll
  for (int p = 1; p < 6; ++p) s += part[p][e];
  if (e >= 36) {  // rhs of pose j
    const int r = e - 36;
    const double val = pre - s;
    d.Spk[(size_t)d.B * 36 + (size_t)j * 6 + r] = val;
    if (DIRECT)  // rhs rides as row `npad` of the dense matrix (see k_scatter)
      d.L[(size_t)(pcj + r) * d.ld + d.npad] = val;
    return;
  }
  const double val = (j == k) ? (pre - s) : -s;
  d.Spk[(size_t)blk * 36 + e] = val;
  if (DIRECT) {  // same placement as k_scatter
    const int r = e / 6, c = e % 6;
    int row = pck + c, col = pcj + r;
    if (j == k && row < col) return;
    if (row < col) {
      const int t2 = row;
      row = col;
      col = t2;
    }
    d.L[(size_t)col * d.ld + row] = val;
  }
}
__global__ __launch_bounds__(kBlock) void k_schur_final(DevProblem d) {
  __shared__ double part[6][42];
  schur_final_body<false>(d, blockIdx.x, part);
}
// Single GPU: the same, and every value is also placed in the dense matrix
// (the packed buffer is still written: the readers use it).
__global__ __launch_bounds__(kBlock) void k_schur_final_direct(DevProblem d) {
  __shared__ double part[6][42];
  schur_final_body<true>(d, blockIdx.x, part);
}

// Packed S blocks and rhs -> dense column-major lower matrix (reference
// :892-902), after the multi-GPU all-reduce of the packed buffer.
__global__ __launch_bounds__(kBlock) void k_scatter(DevProblem d) {
  if (d.ctrl->done) return;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t nb = d.B * 36;
  if (t >= nb + 6 * (int64_t)d.N) return;
  const double val = d.Spk[t];
  if (t >= nb) {  // rhs rides as row `npad`
    const int e = (int)(t - nb);
    d.L[(size_t)(d.pose_col[e / 6] + e % 6) * d.ld + d.npad] = val;
    return;
  }
  const int64_t blk = t / 36;
  const int e = (int)(t - blk * 36);
  const int r = e / 6, c = e % 6;
  const int j = d.sblk_j[blk], k = d.sblk_k[blk];
  // S_jk[r][c] lives at dense (row, col) = (col_of(k)+c, col_of(j)+r) or its
  // transpose, whichever is in the LOWER triangle under the tile ordering
  int row = d.pose_col[k] + c, col = d.pose_col[j] + r;
  // diagonal block: entries (r,c) and (c,r) differ in the last bits (V W^T is
  // not bitwise symmetric) -> only the lower one is stored, never both
  if (j == k && row < col) return;
  if (row < col) {
    const int t2 = row;
    row = col;
    col = t2;
  }
  d.L[(size_t)col * d.ld + row] = val;
}

// se3 exponential (reference :1046-1082) composed onto T_jw (:487-494),
// pose-side model terms (:437-441) and sum |x_j| (:962).  Block b writes its
// partial sums to pose_part[2 + 2b ..]; k_scalars adds them in block order.
// Runs as the first kPoseGrid workgroups of the k_backsub_update launch (it needs
// x only, like the back-substitution): no launch, no stream fork of its own.
// (as in ba_dense_tile.inc: the in-launch hand-offs of this file — k_backsub_lin — rely on gfx9-
//  family behaviour: agent-scope relaxed atomics compile to sc1 accesses, s_waitcnt vmcnt(0)
//  also drains the wave's stores)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "ba_kernels.hip relies on gfx9-family cache behaviour (sc1 hand-offs, LDS-only barriers): build for gfx950"
#endif
// Hand-offs between workgroups of ONE launch (k_backsub_lin): sc1 stores reach memory past the
// XCD's L2, sc1 loads miss it — no fences (an agent-scope release / acquire writes back /
// invalidates the whole L2 of the XCD: 2 000 of them per launch made it 3 times slower).
template <bool SC1>
__device__ __forceinline__ void st_sc1(double *p, double v) {
  if (SC1)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    *p = v;
}
// (SC1 — k_backsub_lin: the new poses are read by workgroups of the same launch on other XCDs,
//  whose L2 is a different one: they leave with sc1 stores)
template <bool SC1 = false>
__device__ __forceinline__ void pose_update_body(const DevProblem &d, const int bid, double *sm) {
  const int cur = d.ctrl->cur;
  const int lbp = d.ctrl->lcur;
  const double lp1p = 1.0 + d.ctrl->lambda;
  if (bid == 0 && threadIdx.x == 0) {  // buffers of this iteration's trial point
    d.ctrl->tcur = cur ^ 1;
    d.ctrl->tlcur = lbp ^ 1;
  }
  const double *__restrict__ Tc = d.poses[cur];
  double *__restrict__ Tt = d.poses[cur ^ 1];
  double est = 0.0, nrm = 0.0;
  for (int j = bid * kBlock + threadIdx.x; j < d.N; j += kPoseGrid * kBlock) {
    const double *xj = d.x + (size_t)j * 6;
    const double v0 = xj[0], v1 = xj[1], v2 = xj[2];
    const double w0 = xj[3], w1 = xj[4], w2 = xj[5];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double wx[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double wx2[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        wx2[r * 3 + c] = wx[r * 3 + 0] * wx[0 * 3 + c] +
                         wx[r * 3 + 1] * wx[1 * 3 + c] +
                         wx[r * 3 + 2] * wx[2 * 3 + c];
    double ca, cb, va, vb;
    if (theta < 1e-7) {
      ca = 1.0;
      cb = 0.5;
      va = 0.5;
      vb = 0.33333333333333333333333333;
    } else {
      const double st = sin(theta), ct = cos(theta);
      ca = st / theta;
      cb = (1.0 - ct) / (theta * theta);
      va = cb;
      vb = (theta - st) / (theta * theta * theta);
    }
    double dR[9], V[9], dt[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double id = (k % 4 == 0) ? 1.0 : 0.0;
      dR[k] = id + ca * wx[k] + cb * wx2[k];
      V[k] = id + va * wx[k] + vb * wx2[k];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
      dt[r] = V[r * 3 + 0] * v0 + V[r * 3 + 1] * v1 + V[r * 3 + 2] * v2;
    const double *T = Tc + (size_t)j * 12;
    double *To = Tt + (size_t)j * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        st_sc1<SC1>(&To[r * 3 + c], dR[r * 3 + 0] * T[0 * 3 + c] + dR[r * 3 + 1] * T[1 * 3 + c] +
                                        dR[r * 3 + 2] * T[2 * 3 + c]);
      st_sc1<SC1>(&To[9 + r], dR[r * 3 + 0] * T[9] + dR[r * 3 + 1] * T[10] + dR[r * 3 + 2] * T[11] + dt[r]);
    }
    const double *aj = d.a[lbp] + (size_t)j * 6;
    const double *Aj = d.A[lbp] + (size_t)j * 36;
    double e = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) e += aj[r] * xj[r];
    double q = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double rowc = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r)  // damped A_j (reference :833-844)
        rowc += xj[r] * (r == c ? Aj[r * 6 + c] * lp1p : Aj[r * 6 + c]);
      q += rowc * xj[c];
    }
    est += e + q;
    nrm += sqrt(v0 * v0 + v1 * v1 + v2 * v2 + w0 * w0 + w1 * w1 + w2 * w2);
  }
  const double t0 = block_sum(est, sm);
  const double t1 = block_sum(nrm, sm);
  if (threadIdx.x == 0) {
    d.pose_part[2 + 2 * bid + 0] = t0;
    d.pose_part[2 + 2 * bid + 1] = t1;
  }
}

// y_i, trial point, landmark-side model terms and |y_i|.
// y_i = Cinv_i b_i - Cinv_i (sum_j B_ji^T x_j)  (reference :910-917; CinvBt is
// never materialised) — the same vector sum_j B_ji^T x_j is the cross term of
// the quadratic model (reference :447-452), so W is read once.
// One workgroup per chunk of consecutive landmarks (<= kSchurPairs pairs per
// tile): the compact W records are copied to LDS with contiguous 16-byte loads,
// lane (pair, c) forms u_c = sum_r W[r][c] x_j[r], then one thread per landmark sums
// its pairs and finishes y, the trial point and the model terms.
// Dependent-load chain per workgroup: chunk record -> {W, pose indices, the
// landmarks' own data} -> x_j gather; everything of one level is issued together.
#define BACKSUB_ISSUE(t0_, np_)                                                 \
  {                                                                             \
    const double2 *src_ = (const double2 *)(Wg + (size_t)(t0_) * kWStride);     \
    _Pragma("unroll") for (int k_ = 0; k_ < kBsRW; ++k_) {                      \
      const int t_ = tid + k_ * kBlock;                                         \
      rw[k_] = (t_ < (np_) * 6) ? src_[t_] : make_double2(0.0, 0.0);            \
    }                                                                           \
    _Pragma("unroll") for (int k_ = 0; k_ < kBsRX; ++k_) {                      \
      const int t_ = tid + k_ * kBlock;                                         \
      rpj[k_] = (t_ < (np_) * 3) ? d.pair_pose[(t0_) + t_ / 3] : 0;             \
    }                                                                           \
  }
// x_j gather of the tile whose pose indices are in rpj: lane (pair, c) fetches
// x_j[2c], x_j[2c+1]
#define BACKSUB_GATHER(np_)                                                     \
  {                                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < kBsRX; ++q_) {                      \
      const int t_ = tid + q_ * kBlock;                                         \
      const int c_ = t_ - (t_ / 3) * 3;                                         \
      rx[q_] = (t_ < (np_) * 3)                                                 \
                   ? *(const double2 *)(d.x + (size_t)rpj[q_] * 6 + 2 * c_)     \
                   : make_double2(0.0, 0.0);                                    \
    }                                                                           \
  }
constexpr int kBsRW = (kSchurPairs * (kWStride / 2) + kBlock - 1) / kBlock;
constexpr int kBsRX = (kSchurPairs * 3 + kBlock - 1) / kBlock;

// Each workgroup handles kBsChunks consecutive chunks: while chunk k goes
// through its LDS phases, the W blocks and pose indices of chunk k+1 are already
// in flight (into the registers chunk k has just released), so only the first
// chunk of a workgroup pays the full load latency.
#ifdef BA_BS_DBG
__device__ long long g_bs_dbg[96];
#define BS_STAMP() { if (bs_on && bs_n < 96) bs_s[bs_n++] = clock64(); }
#else
#define BS_STAMP()
#endif
// LDS of the landmark roles of k_backsub_update (doubles): chunk role = W image,
// x_j image, u; group role = per wave a W image of <= 64 pairs and their u
constexpr int kBgLm = 16;  // landmarks per wave step of the group role, at most
constexpr int kBgArea = 64 * kWStride + 64 * 3 + 64 * 12 + 64 * 3;  // W image, u, landmark data and sums of a block
constexpr int kBsLds = (kSchurPairs * (kWStride + 6 + 3) > 4 * kBgArea) ? kSchurPairs * (kWStride + 6 + 3) : 4 * kBgArea;

// Chunk role: bx = index of the workgroup among the chunk workgroups; it handles
// the chunks lin_chunk0 + bx * kBsChunks ...; part = its entry of lm_part.
__device__ __forceinline__ void backsub_chunk_body(const DevProblem &d, const int bx, const int part, double *lds,
                                                   double *sm, int *recs) {
  double *Ws = lds;                          // kSchurPairs * kWStride
  double *Xs = Ws + kSchurPairs * kWStride;  // kSchurPairs * 6
  double *Us = Xs + kSchurPairs * 6;         // kSchurPairs * 3
  const int tid = threadIdx.x;
#ifdef BA_BS_DBG
  __shared__ long long bs_s[96];
  const bool bs_on = bx == 2000 && threadIdx.x == 0;
  int bs_n = 0;
#endif
  BS_STAMP()
  const int c0 = d.lin_chunk0 + bx * kBsChunks;
  const int nk = min(kBsChunks, d.n_bchunk - c0);
  // the workgroup's chunk records -> LDS (vector loads), first one also scalar
  const DevProblem::LmChunk lc0 = d.lm_chunk[c0];
  if (tid < nk * 8) recs[tid] = ((const int *)d.lm_chunk)[(size_t)c0 * 8 + tid];
  const int done = d.ctrl->done;
  const int cur = d.ctrl->cur;
  const int lbs = d.ctrl->lcur;
  const double lp1 = 1.0 + d.ctrl->lambda;
  BA_KEEP_S((int)lc0.pb);
  BA_KEEP_S(lc0.np);
  BA_KEEP_S(done);
  if (done) return;
  const double *__restrict__ Xc = d.pts[cur];
  double *__restrict__ Xt = d.pts[cur ^ 1];
  const double *__restrict__ Wg = d.W[lbs];
  const double *__restrict__ bg = d.b[lbs];
  const double *__restrict__ Cg = d.Cu[lbs];
  double2 rw[kBsRW];
  int rpj[kBsRX];
  // record of the chunk whose W is in the registers
  int64_t r_pb = lc0.pb;
  int r_l0 = lc0.l0, r_nl = lc0.nl, r_np = lc0.np;
  double2 rx[kBsRX];
  {
    const int np0 = r_np < kSchurPairs ? r_np : kSchurPairs;
    BACKSUB_ISSUE(r_pb, np0)
    BACKSUB_GATHER(np0)
  }
  __syncthreads();  // recs
  BS_STAMP()
  double est_acc = 0.0, nrm_acc = 0.0;
  for (int k = 0; k < nk; ++k) {
    const int64_t pb = r_pb, pe = r_pb + r_np;
    const int l0 = r_l0, nl = r_nl;
    // the landmark owned by this thread (if any): its data is needed last but
    // depends only on the chunk record, so it is requested now
    const int i = l0 + tid;
    const bool own = tid < nl;
    int64_t q0 = 0, q1 = 0;
    double ci[6], Xi[3], bi[3], C[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) ci[e] = C[e] = 0.0;
#pragma unroll
    for (int e = 0; e < 3; ++e) Xi[e] = bi[e] = 0.0;
    if (own) {
      q0 = d.lm_pair_ptr[i];
      q1 = d.lm_pair_ptr[i + 1];
#pragma unroll
      for (int e = 0; e < 3; ++e) Xi[e] = Xc[(size_t)i * 3 + e];
#pragma unroll
      for (int e = 0; e < 3; ++e) bi[e] = bg[(size_t)i * 3 + e];
#pragma unroll
      for (int e = 0; e < 6; ++e) C[e] = Cg[(size_t)i * 6 + e];
    }
    BS_STAMP()
    double bx0 = 0, bx1 = 0, bx2 = 0;  // sum_j W_ji^T x_j
    int64_t t0 = pb;
    int np = (int)min((int64_t)kSchurPairs, pe - t0);
    for (;;) {
      // (rx: the x_j of this tile, gathered one step ahead)
      {
        double2 *dst = (double2 *)Ws;
#pragma unroll
        for (int q = 0; q < kBsRW; ++q) {
          const int t = tid + q * kBlock;
          if (t < np * 6) dst[t] = rw[q];
        }
        double2 *xd = (double2 *)Xs;
#pragma unroll
        for (int q = 0; q < kBsRX; ++q) {
          const int t = tid + q * kBlock;
          if (t < np * 3) xd[t] = rx[q];
        }
      }
      BS_STAMP()
      const bool last_tile = t0 + kSchurPairs >= pe;
      if (last_tile && k + 1 < nk) {  // next chunk's loads fly during this one's phases
        const int *rr = recs + (k + 1) * 8;
        r_pb = uni64(*(const int64_t *)rr);
        r_l0 = __builtin_amdgcn_readfirstlane(rr[4]);
        r_nl = __builtin_amdgcn_readfirstlane(rr[5]);
        r_np = __builtin_amdgcn_readfirstlane(rr[6]);
        const int npn = r_np < kSchurPairs ? r_np : kSchurPairs;
        BACKSUB_ISSUE(r_pb, npn)
      }
      BS_STAMP()
      __syncthreads();
      BS_STAMP()
      for (int t = tid; t < np * 3; t += kBlock) {
        const int lp = t / 3, c = t - lp * 3;
        // u_c = sum_r B_ji[r][c] x_j[r] = K[:,c] . (x_t + x_r x X_ij)
        const double *xj = Xs + lp * 6;
        const double *w = Ws + lp * kWStride;
        const double X0 = w[9], X1 = w[10], X2 = w[11];
        const double e0 = xj[0] + (xj[4] * X2 - xj[5] * X1);
        const double e1 = xj[1] + (xj[5] * X0 - xj[3] * X2);
        const double e2 = xj[2] + (xj[3] * X1 - xj[4] * X0);
        Us[t] = fma(w[6 + c], e2, fma(w[3 + c], e1, w[c] * e0));
      }
      BS_STAMP()
      __syncthreads();
      BS_STAMP()
      if (own) {
        const int64_t a = max(q0, t0), b = min(q1, t0 + np);
        for (int64_t p = a; p < b; ++p) {
          const double *u = Us + (p - t0) * 3;
          bx0 += u[0];
          bx1 += u[1];
          bx2 += u[2];
        }
      }
      BS_STAMP()
      if (last_tile) break;  // uniform
      t0 += kSchurPairs;
      np = (int)min((int64_t)kSchurPairs, pe - t0);
      __syncthreads();  // LDS is rewritten
      BACKSUB_ISSUE(t0, np)
      BACKSUB_GATHER(np)
    }
    if (k + 1 < nk) {  // the next chunk's pose indices have arrived: its x_j gather
      const int npn = r_np < kSchurPairs ? r_np : kSchurPairs;
      BACKSUB_GATHER(npn)
    }
    double est = 0.0, nrm = 0.0;
    if (own) {
      // damping and inverse (reference :846-856), then Cinv_i b_i (:855): formed
      // here from the undamped C_i instead of being read
      {
        const double cdm[6] = {C[0] * lp1, C[1], C[2], C[3] * lp1, C[4], C[5] * lp1};
        ldlt3_inverse(cdm, ci);  // (the fast path of spd3_inverse costs this loop its registers)
      }
      const double cb0 = ci[0] * bi[0] + ci[1] * bi[1] + ci[2] * bi[2];
      const double cb1 = ci[1] * bi[0] + ci[3] * bi[1] + ci[4] * bi[2];
      const double cb2 = ci[2] * bi[0] + ci[4] * bi[1] + ci[5] * bi[2];
      const double y0 = cb0 - (ci[0] * bx0 + ci[1] * bx1 + ci[2] * bx2);
      const double y1 = cb1 - (ci[1] * bx0 + ci[3] * bx1 + ci[4] * bx2);
      const double y2 = cb2 - (ci[2] * bx0 + ci[4] * bx1 + ci[5] * bx2);
      double *yo = d.y + (size_t)i * 3;
      yo[0] = y0; yo[1] = y1; yo[2] = y2;
      double *Xo = Xt + (size_t)i * 3;
      Xo[0] = Xi[0] + y0;
      Xo[1] = Xi[1] + y1;
      Xo[2] = Xi[2] + y2;
      // reference :443-452 with the damped C_i (diagonal times 1 + lambda, :846-852)
      const double C0 = C[0] * lp1, C3 = C[3] * lp1, C5 = C[5] * lp1;
      double e = bi[0] * y0 + bi[1] * y1 + bi[2] * y2;
      const double r0 = y0 * C0 + y1 * C[1] + y2 * C[2];
      const double r1 = y0 * C[1] + y1 * C3 + y2 * C[4];
      const double r2 = y0 * C[2] + y1 * C[4] + y2 * C5;
      e += r0 * y0 + r1 * y1 + r2 * y2;
      e += 2.0 * (y0 * bx0 + y1 * bx1 + y2 * bx2);
      est = e;
      nrm = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
    }
    BS_STAMP()
    est_acc += est;  // a thread owns at most one landmark per chunk: fixed order
    nrm_acc += nrm;
  }
  // one pair of partial sums per workgroup (k_scalars adds them in block order)
  block_sum2(est_acc, nrm_acc, sm);
  if (tid == 0) {
    d.lm_part[2 * part + 0] = est_acc;
    d.lm_part[2 * part + 1] = nrm_acc;
  }
  BS_STAMP()
#ifdef BA_BS_DBG
  if (bs_on) for (int q = 0; q < 96; ++q) g_bs_dbg[q] = q < bs_n ? bs_s[q] : 0;
#endif
}

// Group role: the landmarks of one k_lin_grp piece (ba_plan.h LinDesc: one
// covisibility group, every landmark seen by the same d free poses, pair (il, jj) =
// p0 + d il + jj).  The four waves are independent; a wave step covers 64 / d
// consecutive landmarks, lane = (landmark, pose jj of the group): x_j is a LANE
// CONSTANT (registers, no gather, no pose indices), the step's W records are one
// contiguous range, read with consecutive 16-byte pieces two steps ahead and
// redistributed through the wave's LDS image.  u = B_ji^T x_j per lane; one lane per
// landmark adds its d vectors in pair order, inverts the damped C_i and updates the
// point (reference :846-856, :907-925, :443-452).
template <bool FUSED = false>
__device__ __forceinline__ void backsub_grp_body(const DevProblem &d, const int piece, const int part, double *lds,
                                                 double *sm, int *bl_flag = nullptr, int gen = 0) {
  const DevProblem::LinDesc *gp = d.lin_desc + piece;
  const int64_t p0 = gp->p0;
  const int l0 = gp->l0, nl = gp->nl, dd = gp->d;
  const int done = d.ctrl->done;
  const int cur = d.ctrl->cur;
  const int lbs = d.ctrl->lcur;
  const double lp1 = 1.0 + d.ctrl->lambda;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // (dd == 0: landmarks seen by fixed poses only: no pair, y_i = Cinv_i b_i)
  const int ddv = dd > 0 ? dd : 1;
  const int nlwb = 64 / ddv < kBgLm ? 64 / ddv : kBgLm;  // landmarks per wave step
  const int ilw = lane / ddv, jj = lane - ilw * ddv;
  const bool lane_on = dd > 0 && ilw < nlwb;
  const int pose = dd > 0 ? d.pair_pose[p0 + jj] : 0;
  if (done) return;
  double xj[6];
  {
    const double2 *xp = (const double2 *)(d.x + (size_t)pose * 6);
    const double2 a0 = xp[0], a1 = xp[1], a2 = xp[2];
    xj[0] = a0.x; xj[1] = a0.y; xj[2] = a1.x; xj[3] = a1.y; xj[4] = a2.x; xj[5] = a2.y;
  }
  double *__restrict__ Xt = d.pts[cur ^ 1];
  const double2 *__restrict__ Wg2 = (const double2 *)d.W[lbs];
  double *Wim = lds + wv * kBgArea;  // 64 * kWStride: W image of the step
  double *Us = Wim + 64 * kWStride;  // 64 * 3: u per pair
  double *Lm = Us + 64 * 3;          // per step of a block: C_i (6), b_i (3), X_i (3) of its landmarks
  double *Bx = Lm + 64 * 12;         // 64 * 3: sum_j W_ji^T x_j of the block's landmarks
  // The landmark arithmetic (3x3 inverse, y_i, model terms: ~250 instructions) runs
  // once per BLOCK of S = 64 / nlwb steps with one lane per landmark (all lanes
  // busy), not once per step with nlwb of 64 lanes busy.
  const int S = 64 / nlwb;
  const int per_step = 4 * nlwb;
  const int nstep = (nl + per_step - 1) / per_step;
  auto il0_of = [&](int st) { return (st * 4 + wv) * nlwb; };
  // The step's landmark data = three contiguous ranges (C_i: 6 nlwb doubles, b_i and
  // X_i: 3 nlwb each), fetched as 12 nlwb <= 192 doubles, three per lane: entry q =
  // lane + 64 k of the list {C, b, X} is element lm_e[k] of landmark lm_l[k] of array
  // lm_p[k] (stride lm_m[k] doubles per landmark); all lane constants.
  const double *lm_p[3];
  int lm_l[3], lm_e[3], lm_m[3], lm_q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int q = min(lane + 64 * k, nlwb * 12 - 1);
    const int nc = nlwb * 6, nb = nlwb * 9;
    const int a = q < nc ? 0 : (q < nb ? 1 : 2);
    const int r = q - (a == 0 ? 0 : (a == 1 ? nc : nb));
    lm_m[k] = a == 0 ? 6 : 3;
    lm_l[k] = r / lm_m[k];
    lm_e[k] = r - lm_l[k] * lm_m[k];
    lm_p[k] = (a == 0 ? d.Cu[lbs] : (a == 1 ? d.b[lbs] : d.pts[cur])) + (size_t)l0 * lm_m[k] + lm_e[k];
    lm_q[k] = q;
  }
  // requests of step ST: its W range (consecutive 16-byte pieces; past the end the
  // last piece again) and its landmark data (clamped indices: no conditional loads)
  // (named scalars, not arrays: a register ring declared as an array ended up in
  //  scratch memory here)
  double2 rwA0, rwA1, rwA2, rwA3, rwA4, rwA5, rwB0, rwB1, rwB2, rwB3, rwB4, rwB5;
  double lmA0, lmA1, lmA2, lmB0, lmB1, lmB2;
#define BSG_ISSUE(R, L, ST)                                                         \
  {                                                                                 \
    const int il0_ = il0_of(ST);                                                    \
    const int nls_ = max(0, min(nlwb, nl - il0_));                                  \
    const int last_ = max(0, nls_ * dd * 6 - 1);                                    \
    const int ilb_ = nls_ > 0 ? il0_ : 0;                                           \
    const int lml_ = max(0, nls_ - 1);                                              \
    const double2 *src_ = Wg2 + (dd > 0 ? (size_t)(p0 + (int64_t)ilb_ * dd) * 6 : (size_t)0); \
    R##0 = src_[min(lane, last_)];                                                  \
    R##1 = src_[min(lane + 64, last_)];                                             \
    R##2 = src_[min(lane + 128, last_)];                                            \
    R##3 = src_[min(lane + 192, last_)];                                            \
    R##4 = src_[min(lane + 256, last_)];                                            \
    R##5 = src_[min(lane + 320, last_)];                                            \
    L##0 = lm_p[0][(size_t)(ilb_ + min(lm_l[0], lml_)) * lm_m[0]];                  \
    L##1 = lm_p[1][(size_t)(ilb_ + min(lm_l[1], lml_)) * lm_m[1]];                  \
    L##2 = lm_p[2][(size_t)(ilb_ + min(lm_l[2], lml_)) * lm_m[2]];                  \
  }
#define BSG_WAVE_SYNC()                                                             \
  {                                                                                 \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                          \
    __builtin_amdgcn_wave_barrier();                                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                          \
  }
  double est_acc = 0.0, nrm_acc = 0.0;
  double *dump = d.lin_dump + (size_t)((blockIdx.x & 63) * 4 + wv) * 4;
  int sb = 0, st_b0 = 0;  // step inside the block, first step of the block
  const int ls_ = lane / nlwb, lli = lane - ls_ * nlwb;  // landmark phase: lane -> (step of the block, landmark)
  const int sli = lane / 3, slc = lane - sli * 3;       // sum phase: lane -> (landmark of the step, coordinate)
#define BSG_STEP(R, L, TAG)                                                         \
  {                                                                                 \
    /* (the two instances must stay two code paths) */                              \
    asm volatile("; k_backsub_update group step " TAG);                             \
    /* registers -> the wave's LDS images (linear) */                               \
    ((double2 *)Wim)[lane] = R##0;                                                  \
    ((double2 *)Wim)[lane + 64] = R##1;                                             \
    ((double2 *)Wim)[lane + 128] = R##2;                                            \
    ((double2 *)Wim)[lane + 192] = R##3;                                            \
    ((double2 *)Wim)[lane + 256] = R##4;                                            \
    ((double2 *)Wim)[lane + 320] = R##5;                                            \
    Lm[sb * nlwb * 12 + lm_q[0]] = L##0;                                            \
    Lm[sb * nlwb * 12 + lm_q[1]] = L##1;                                            \
    Lm[sb * nlwb * 12 + lm_q[2]] = L##2;                                            \
    BSG_ISSUE(R, L, st + 2)                                                         \
    BSG_WAVE_SYNC()                                                                 \
    if (lane_on) {                                                                  \
      /* u_c = sum_r B_ji[r][c] x_j[r] = K[:,c] . (x_t + x_r x X_ij) */             \
      const double2 *w2 = (const double2 *)(Wim + lane * kWStride);                 \
      const double2 k01 = w2[0], k23 = w2[1], k45 = w2[2], k67 = w2[3], k8x = w2[4], xx = w2[5]; \
      const double X0 = k8x.y, X1 = xx.x, X2 = xx.y;                                \
      const double e0 = xj[0] + (xj[4] * X2 - xj[5] * X1);                          \
      const double e1 = xj[1] + (xj[5] * X0 - xj[3] * X2);                          \
      const double e2 = xj[2] + (xj[3] * X1 - xj[4] * X0);                          \
      double *u = Us + lane * 3;                                                    \
      u[0] = fma(k67.x, e2, fma(k23.y, e1, k01.x * e0));                            \
      u[1] = fma(k67.y, e2, fma(k45.x, e1, k01.y * e0));                            \
      u[2] = fma(k8x.x, e2, fma(k45.y, e1, k23.x * e0));                            \
    }                                                                               \
    BSG_WAVE_SYNC()                                                                 \
    /* sums of the step's landmarks, pair order: lane (landmark, coordinate) */     \
    if (lane < nlwb * 3) {                                                          \
      const double *u = Us + sli * dd * 3 + slc;                                    \
      double bx_ = 0.0;                                                             \
      for (int q_ = 0; q_ < dd; ++q_) bx_ += u[q_ * 3];                             \
      Bx[(sb * nlwb + sli) * 3 + slc] = bx_;                                        \
    }                                                                               \
    BSG_WAVE_SYNC()                                                                 \
    /* landmark arithmetic at the end of a block (uniform branch: LDS and arithmetic */ \
    /* only); the stores follow unconditionally — lanes and steps with nothing to  */ \
    /* store write zeros to a dump: a store under a branch makes the compiler's    */ \
    /* vmcnt bookkeeping wait for all loads in flight (one memory latency per step) */ \
    const bool do_lm_ = (sb + 1 == S) || (st + 1 >= nstep);                         \
    double y0 = 0.0, y1 = 0.0, y2 = 0.0, xn0 = 0.0, xn1 = 0.0, xn2 = 0.0;           \
    bool lm_on_ = false;                                                            \
    size_t i_ = 0;                                                                  \
    if (do_lm_) {                                                                   \
      const int il_ = il0_of(st_b0 + ls_) + lli;                                    \
      lm_on_ = ls_ <= sb && il_ < nl;                                               \
      i_ = (size_t)l0 + il_;                                                        \
      const double *lmp = Lm + ls_ * nlwb * 12;                                     \
      double C[6], bi[3], Xi[3];                                                    \
      _Pragma("unroll") for (int e_ = 0; e_ < 6; ++e_) C[e_] = lmp[lli * 6 + e_];   \
      _Pragma("unroll") for (int e_ = 0; e_ < 3; ++e_) bi[e_] = lmp[nlwb * 6 + lli * 3 + e_]; \
      _Pragma("unroll") for (int e_ = 0; e_ < 3; ++e_) Xi[e_] = lmp[nlwb * 9 + lli * 3 + e_]; \
      const double bx0 = Bx[lane * 3], bx1 = Bx[lane * 3 + 1], bx2 = Bx[lane * 3 + 2]; \
      double ci[6];                                                                 \
      {                                                                             \
        const double cdm[6] = {C[0] * lp1, C[1], C[2], C[3] * lp1, C[4], C[5] * lp1}; \
        spd3_inverse(cdm, ci);                                                      \
      }                                                                             \
      const double cb0 = ci[0] * bi[0] + ci[1] * bi[1] + ci[2] * bi[2];             \
      const double cb1 = ci[1] * bi[0] + ci[3] * bi[1] + ci[4] * bi[2];             \
      const double cb2 = ci[2] * bi[0] + ci[4] * bi[1] + ci[5] * bi[2];             \
      y0 = cb0 - (ci[0] * bx0 + ci[1] * bx1 + ci[2] * bx2);                         \
      y1 = cb1 - (ci[1] * bx0 + ci[3] * bx1 + ci[4] * bx2);                         \
      y2 = cb2 - (ci[2] * bx0 + ci[4] * bx1 + ci[5] * bx2);                         \
      xn0 = Xi[0] + y0;                                                             \
      xn1 = Xi[1] + y1;                                                             \
      xn2 = Xi[2] + y2;                                                             \
      /* reference :443-452 with the damped C_i (diagonal times 1 + lambda) */      \
      const double C0 = C[0] * lp1, C3 = C[3] * lp1, C5 = C[5] * lp1;               \
      double e = bi[0] * y0 + bi[1] * y1 + bi[2] * y2;                              \
      const double r0 = y0 * C0 + y1 * C[1] + y2 * C[2];                            \
      const double r1 = y0 * C[1] + y1 * C3 + y2 * C[4];                            \
      const double r2 = y0 * C[2] + y1 * C[4] + y2 * C5;                            \
      e += r0 * y0 + r1 * y1 + r2 * y2;                                             \
      e += 2.0 * (y0 * bx0 + y1 * bx1 + y2 * bx2);                                  \
      est_acc += lm_on_ ? e : 0.0;                                                  \
      nrm_acc += lm_on_ ? sqrt(y0 * y0 + y1 * y1 + y2 * y2) : 0.0;                  \
    }                                                                               \
    {                                                                               \
      double *yo = lm_on_ ? d.y + i_ * 3 : dump;                                    \
      yo[0] = y0; yo[1] = y1; yo[2] = y2;                                           \
      double *Xo = lm_on_ ? Xt + i_ * 3 : dump;                                     \
      Xo[0] = xn0; Xo[1] = xn1; Xo[2] = xn2;                                        \
    }                                                                               \
    if (do_lm_) {                                                                   \
      sb = 0;                                                                       \
      st_b0 = st + 1;                                                               \
    } else {                                                                        \
      ++sb;                                                                         \
    }                                                                               \
    BSG_WAVE_SYNC()                                                                 \
  }
  BSG_ISSUE(rwA, lmA, 0)
  BSG_ISSUE(rwB, lmB, 1)
  int st = 0;
  while (st < nstep) {
    BSG_STEP(rwA, lmA, "A")
    if (++st >= nstep) break;
    BSG_STEP(rwB, lmB, "B")
    ++st;
  }
#undef BSG_STEP
#undef BSG_ISSUE
#undef BSG_WAVE_SYNC
  block_sum2(est_acc, nrm_acc, sm);
  if (tid == 0) {
    d.lm_part[2 * part + 0] = est_acc;
    d.lm_part[2 * part + 1] = nrm_acc;
  }
  if (FUSED) {
    // k_backsub_lin: the piece's trial points go out ONCE MORE with sc1 stores (past this XCD's
    // L2, for the linearisation workgroups on the other XCDs), then its flag is raised.  (sc1
    // on the step loop's own stores would also hit the lanes that write to the dump word:
    // thousands of write-throughs to one address, the role 4 times slower.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = tid; e < nl * 3; e += kBlock) st_sc1<true>(&Xt[(size_t)l0 * 3 + e], Xt[(size_t)l0 * 3 + e]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&bl_flag[piece], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Workgroup roles (by blockIdx, wave-uniform): [0, kPoseGrid) the pose update (see
// pose_update_body); then one workgroup per covisibility-group piece (n_bs_grp of
// them); then the chunks behind the groups, kBsChunks per workgroup.  Entry `part`
// of lm_part = blockIdx - kPoseGrid.
// GROUPS = false: problems without covisibility groups (chunk role only): the LDS of
// the chunk role alone and four workgroups per CU instead of two.
template <bool GROUPS>
__global__ __launch_bounds__(kBlock, GROUPS ? 2 : 4) void k_backsub_update(DevProblem d) {
  __shared__ __attribute__((aligned(16))) double lds[GROUPS ? kBsLds : kSchurPairs * (kWStride + 6 + 3)];
  __shared__ double sm[8];
  __shared__ int recs[kBsChunks * 8];
  if (blockIdx.x < kPoseGrid) {
    if (d.ctrl->done) return;
    pose_update_body(d, blockIdx.x, sm);
    return;
  }
  const int part = blockIdx.x - kPoseGrid;
  if (part >= d.n_lm_part) {  // role: reset of the factor tiles for the NEXT reduced system (k_dense_init)
    if (d.ctrl->done) return;
    const int z = part - d.n_lm_part;
    const int I = d.zt_I[z], J = d.zt_J[z], nb = d.nb;
    for (int e = threadIdx.x; e < nb * nb; e += kBlock) {
      const int c = J * nb + e / nb, r = I * nb + e % nb;
      d.L[(size_t)c * d.ld + r] = (r == c && d.col_x[c] < 0) ? 1.0 : 0.0;
    }
    return;
  }
  if (GROUPS && part < d.n_bs_grp)
    backsub_grp_body(d, part, part, lds, sm);
  else
    backsub_chunk_body(d, part - d.n_bs_grp, part, lds, sm, recs);
}
// ---- back-substitution + update AND the trial-point linearisation in ONE launch ----
// (problems whose free landmarks are all in covisibility groups, single GPU; OPT-IN,
// BA_FUSE_BL=1.)  The idea: the two kernels look complementary — k_backsub_update is bound by
// HBM (0.70 of the peak), k_lin_grp by the fp64 VALU (its memory side at 0.44) — and the
// linearisation of a group piece needs only the poses and the points of ITS landmarks, so
// run beside each other on the same CUs the pieces of one might fill the other's idle pipe.
// MEASURED: they do not (see enqueue_iteration): interleaving is slower the finer it is;
// with the roles one after the other the launch equals the two kernels and saves one
// launch gap.  Roles from the block index:
//   [0, kPoseGrid)   pose update; each raises pose_flag[b] = gen when its poses are out;
//   then 2 P blocks: the first K are back-substitution pieces 0..K-1, then linearisation
//                    piece m and back-substitution piece K + m alternate, the last K are
//                    linearisation pieces (K = kBlLead: by the time piece m's linearisation
//                    is dispatched its back-substitution, 2 K - 1 blocks earlier, is over);
//   then             the reset of the factor tiles (k_dense_init role), as in k_backsub_update.
// A linearisation workgroup waits for the pose flags and for its piece's flag — raised by
// a block with a smaller index, i.e. one dispatched earlier; bounded polls — and reads poses and
// points with sc1 loads (lin_grp_body<FUSED>); the producers store them with sc1 stores and
// drain (s_waitcnt vmcnt(0)) before they raise their flag: no fences (see st_sc1).  Same arithmetic as the two launches
// (BA_FUSE_BL=0), bit for bit.
constexpr int kBlLead = 1 << 24;  // (default: no interleaving — the measured optimum, see enqueue_iteration)
// (MASKED: the launch holds pieces of superset groups; their variant of the linearisation
//  then takes every piece — for a piece without padded slots it performs the plain
//  variant's arithmetic, the last writer found from the ballot is the static one)
template <bool LDSCAM, bool MASKED>
__global__ __launch_bounds__(kBlock, 2) void k_backsub_lin(DevProblem d, int gen, int *bad, int lead) {
  __shared__ __attribute__((aligned(16))) double lds[kBsLds];
  static_assert(kBsLds >= 4 * kLgArea + kLinGrpSteps * 4 * 7 * 3, "the linearisation role's LDS areas and its piece's points fit");
  __shared__ double cams_s[kCamLds * 16];
  __shared__ double sm[8];
  __shared__ int slot_b[kGrpMaxPoses], slot_e[kGrpMaxPoses];
  if (blockIdx.x < kPoseGrid) {
    if (d.ctrl->done) return;
    pose_update_body<true>(d, blockIdx.x, sm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&d.pose_flag[blockIdx.x], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int P = d.n_bs_grp;
  const int K = P < lead ? P : lead;
  const int q = blockIdx.x - kPoseGrid;
  // the alternating region in runs of eight: workgroups go to the eight XCDs round-robin, a
  // strict alternation would put every linearisation piece on an even XCD and every back-
  // substitution piece on an odd one
  const int Q = P - K, G = (Q + 7) >> 3;
  const int n_roles = 2 * K + 16 * G;
  if (q >= n_roles) {  // role: reset of the factor tiles for the NEXT reduced system (k_dense_init)
    if (d.ctrl->done) return;
    const int z = q - n_roles;
    const int I = d.zt_I[z], J = d.zt_J[z], nb = d.nb;
    for (int e = threadIdx.x; e < nb * nb; e += kBlock) {
      const int c = J * nb + e / nb, r = I * nb + e % nb;
      d.L[(size_t)c * d.ld + r] = (r == c && d.col_x[c] < 0) ? 1.0 : 0.0;
    }
    return;
  }
  int piece;
  bool lin;
  if (q < K) {
    piece = q;
    lin = false;
  } else if (q < K + 16 * G) {
    const int r = q - K, g = r >> 4, w = r & 15;
    lin = w < 8;
    const int idx = 8 * g + (w & 7);
    if (idx >= Q) return;  // (padding of the last run)
    piece = lin ? idx : K + idx;
  } else {
    piece = Q + (q - (K + 16 * G));
    lin = true;
  }
  if (!lin) {
    backsub_grp_body<true>(d, piece, piece, lds, sm, d.bl_flag, gen);
  } else
    lin_grp_body<LDSCAM, MASKED, true>(d, 1, piece, lds, cams_s, sm, slot_b, slot_e, d.bl_flag, d.pose_flag, gen, bad);
}

#ifdef BA_BS_DBG
extern "C" int ba_debug_read_bs(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bs_dbg), sizeof(long long) * 96);
}
#endif

// Reduce the block partials into the exchange scalars.
//   mode 0: scal[0] = cost only (initial cost)
//   mode 1: scal[0] = trial cost, scal[1] = model estimate, scal[2] = sum|y|
__device__ void control_step(const DevProblem &d);
__device__ void control_step_vals(const DevProblem &d, double current_cost, double model_est, double sum_y, double sum_x);

// mode 0: cost only; 1: all LM scalars; 2: all LM scalars, then the trust-region
// control step by the same workgroup (single GPU: nothing to all-reduce in between)
// cost_src 0: the k_cost partials (stage API: a cost pass over every observation);
// 1: the partials of k_lin_landmarks (cost as a by-product of the linearisation)
// plus, when fixed landmarks have observations, the k_cost partials of those
constexpr int kScalBlock = 1024;
// fin_sel >= 0: the workgroups behind the first are k_pose_finalize (sel = fin_sel) —
// the pose-side sums of the linearisation that has just finished, in this launch
// instead of on a side stream when nothing else is left for that stream.
__global__ __launch_bounds__(kScalBlock) void k_scalars(DevProblem d, int mode, int cost_src, int fin_sel) {
  if (blockIdx.x > 0) {
    // (no look at ctrl->done: the first workgroup may set it while this one starts, and
    //  A_j, a_j of the last trial point should be complete; summing again is idempotent)
    pose_finalize_body(d, fin_sel, (blockIdx.x - 1) * kScalBlock + threadIdx.x);
    return;
  }
  if (d.ctrl->done) return;
  double c = 0.0, e = 0.0, n = 0.0, pe = 0.0, pn = 0.0;
  if (cost_src == 0 || d.n_obs_lm < d.n_obs) {  // both rounds of the cost partials requested at once
    static_assert(kCostGrid <= 2 * kScalBlock, "two loads per thread");
    const int k1 = threadIdx.x + kScalBlock;
    const double c0 = d.cost_part[threadIdx.x < kCostGrid ? threadIdx.x : 0];
    const double c1 = d.cost_part[k1 < kCostGrid ? k1 : 0];
    c = (threadIdx.x < kCostGrid ? c0 : 0.0) + (k1 < kCostGrid ? c1 : 0.0);
  }
  if (cost_src == 1) {  // eight independent loads in flight per thread
    const double *lp = d.lin_cost_part;
    for (int k0 = threadIdx.x; k0 < d.n_lin_cost; k0 += 8 * kScalBlock) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * kScalBlock;
        v[u] = k < d.n_lin_cost ? lp[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) c += v[u];
    }
  }
  if (mode >= 1) {
    const double2 *lp = (const double2 *)d.lm_part;
    // eight independent loads in flight per thread
    const int n_lp = d.n_lm_part;  // one entry per landmark workgroup of k_backsub_update
    for (int k0 = threadIdx.x; k0 < n_lp; k0 += 8 * kScalBlock) {
      double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * kScalBlock;
        v[u] = k < n_lp ? lp[k] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        e += v[u].x;
        n += v[u].y;
      }
    }
    if (threadIdx.x < kPoseGrid) {
      pe = d.pose_part[2 + 2 * threadIdx.x + 0];
      pn = d.pose_part[2 + 2 * threadIdx.x + 1];
    }
  }
  // five sums, one barrier pair: wave totals -> LDS -> thread 0 adds the 16 waves in order
  __shared__ double sm5[5][kScalBlock / 64];
  {
    const double w0 = wave_sum(c), w1 = wave_sum(e), w2 = wave_sum(n), w3 = wave_sum(pe),
                 w4 = wave_sum(pn);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
      sm5[0][wv] = w0;
      sm5[1][wv] = w1;
      sm5[2][wv] = w2;
      sm5[3][wv] = w3;
      sm5[4][wv] = w4;
    }
  }
  __syncthreads();
  double tc = 0.0, te = 0.0, tn = 0.0, tpe = 0.0, tpn = 0.0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < kScalBlock / 64; ++w) {
      tc += sm5[0][w];
      te += sm5[1][w];
      tn += sm5[2][w];
      tpe += sm5[3][w];
      tpn += sm5[4][w];
    }
  }
  if (threadIdx.x == 0) {
    if (mode >= 1) {
      d.pose_part[0] = tpe;
      d.pose_part[1] = tpn;
    }
    d.scal[0] = tc;
    d.scal[1] = (mode >= 1) ? te + tpe : 0.0;
    d.scal[2] = (mode >= 1) ? tn : 0.0;
    d.scal[3] = 0.0;
    if (mode == 2) control_step_vals(d, tc, (mode >= 1) ? te + tpe : 0.0, (mode >= 1) ? tn : 0.0, tpn);
  }
}

__global__ void k_init_ctrl_cost(DevProblem d) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  d.ctrl->prev_cost = d.scal[0];
  d.ctrl->t_last = wall_clock64();
}

// Trust region, convergence and iteration log (reference :928-1007); one thread.
__device__ void control_step(const DevProblem &d) {
  if (d.ctrl->done) return;
  control_step_vals(d, d.scal[0], d.scal[1], d.scal[2], d.pose_part[1]);
}
// (the reduction workgroup passes the sums in registers: no store -> fence -> load
//  round trip in front of the decision)
__device__ void control_step_vals(const DevProblem &d, const double current_cost, const double model_est,
                                  const double sum_y, const double sum_x) {
  DevCtrl *c = d.ctrl;
  if (c->done) return;
  const double model = -model_est;
  const double previous_cost = c->prev_cost;
  double rho = (current_cost - previous_cost) * 100.0 / model;
  int status;
  double lambda = c->lambda;
  if (c->gn) {
    // plain Gauss-Newton of the refactored solver (reference
    // core/full_bundle_adjustment_solver_refactor.cpp:976-982)
    status = 0;
    rho = 0.0;
    c->cur ^= 1;
    c->lcur ^= 1;
  } else {
    if (rho > 0.25) {
      status = 0;
      c->cur ^= 1;   // the trial buffer becomes the accepted one ...
      c->lcur ^= 1;  // ... and so does the linearisation made at the trial point
    } else {
      status = 2;   // keep the reserved parameters (reference :943)
    }
    if (rho > 0.5) {
      lambda = fmax(1e-10, lambda * c->dec_ratio);
      status = 1;
    } else if (rho <= 0.25) {
      lambda = fmin(100.0, lambda * c->inc_ratio);
    }
  }
  c->lambda = lambda;
  const double n_obs = (double)d.n_obs_global;
  const double average_error = current_cost / n_obs;
  const double cost_change = fabs(current_cost - previous_cost);
  const double total_step = sum_y + sum_x;
  const double avg_step = total_step / (double)(d.N + d.M_global);
  bool conv = (avg_step < c->thr_step) || (cost_change < c->thr_cost);
  if (c->iter >= c->max_iter - 1) conv = false;
  const unsigned long long now = wall_clock64();
  if (c->iter < d.log_cap) {
    DevIterRec &I = d.log[c->iter];
    I.cost = current_cost;
    I.cost_change = cost_change;
    I.average_reprojection_error = average_error;
    I.abs_gradient = 0.0;
    I.abs_step = avg_step;
    I.damping_term = lambda;
    I.iter_time_ms = (double)(now - c->t_last) * 1e-5;  // 100 MHz clock
    I.iteration_status = status;
    I.pad_ = 0;
    I.rho = rho;
    I.model_change = c->gn ? 0.0 : model;
    I.trial_cost = current_cost;
    if (status == 2) {  // reference :995-1000
      I.cost = previous_cost;
      I.cost_change = 0.0;
      I.average_reprojection_error = sqrt(previous_cost / n_obs);
    }
  }
  c->t_last = now;
  c->prev_cost = current_cost;  // even when SKIPPED (reference :1005)
  c->iter += 1;
  c->converged = conv ? 1 : 0;
  if (conv || c->iter >= c->max_iter) c->done = 1;
}

__global__ void k_control(DevProblem d) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  control_step(d);
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace

void launch_cost(const DevProblem &d, int sel, int64_t begin, hipStream_t s) {
  if (begin >= d.n_obs) return;
  const bool lds = d.n_cam <= kCamLds, slim = d.obs_cp != nullptr;
  if (lds && slim) {
    BA_LAUNCH(K_COST, (k_cost<true, true>), dim3(kCostGrid), dim3(kBlock), s, d, sel, begin);
  } else if (lds) {
    BA_LAUNCH(K_COST, (k_cost<true, false>), dim3(kCostGrid), dim3(kBlock), s, d, sel, begin);
  } else if (slim) {
    BA_LAUNCH(K_COST, (k_cost<false, true>), dim3(kCostGrid), dim3(kBlock), s, d, sel, begin);
  } else {
    BA_LAUNCH(K_COST, (k_cost<false, false>), dim3(kCostGrid), dim3(kBlock), s, d, sel, begin);
  }
}

void launch_lin_landmarks(const DevProblem &d, int sel, hipStream_t s) {
  if (d.lin_chunk0 > 0) {  // covisibility groups: landmark and pose side in one pass
    const int n_plain = d.n_lin_plain, n_mask = d.n_lin_desc - d.n_lin_plain;
    if (d.n_cam <= kCamLds) {
      if (n_plain > 0) BA_LAUNCH(K_LIN_GRP, (k_lin_grp<true, false>), dim3(n_plain), dim3(kBlock), s, d, sel, 0);
      if (n_mask > 0) BA_LAUNCH(K_LIN_GRP, (k_lin_grp<true, true>), dim3(n_mask), dim3(kBlock), s, d, sel, n_plain);
    } else {
      if (n_plain > 0) BA_LAUNCH(K_LIN_GRP, (k_lin_grp<false, false>), dim3(n_plain), dim3(kBlock), s, d, sel, 0);
      if (n_mask > 0) BA_LAUNCH(K_LIN_GRP, (k_lin_grp<false, true>), dim3(n_mask), dim3(kBlock), s, d, sel, n_plain);
    }
  }
  const int nch = d.n_bchunk - d.lin_chunk0;
  if (nch <= 0) return;
  if (d.n_cam <= kCamLds)
    BA_LAUNCH(K_LIN_LANDMARKS, k_lin_landmarks<true>, dim3(nch), dim3(kBlock), s, d, sel);
  else
    BA_LAUNCH(K_LIN_LANDMARKS, k_lin_landmarks<false>, dim3(nch), dim3(kBlock), s, d, sel);
}

void launch_lin_poses(const DevProblem &d, int sel, hipStream_t s) {
  if (d.n_achunk > 0) {
    if (d.n_cam <= kCamLds)
      BA_LAUNCH(K_LIN_POSES, k_lin_poses<true>, dim3(cdiv(d.n_achunk, kBlock / 64)), dim3(kBlock), s, d, sel);
    else
      BA_LAUNCH(K_LIN_POSES, k_lin_poses<false>, dim3(cdiv(d.n_achunk, kBlock / 64)), dim3(kBlock), s, d, sel);
  }
  if (d.N > 0)
    BA_LAUNCH(K_POSE_FINALIZE, k_pose_finalize, dim3(cdiv((int64_t)d.N * 27, kBlock)), dim3(kBlock), s, d, sel);
}

void launch_damp_invert(const DevProblem &d, hipStream_t s) {
  // Cinv as an ARRAY is read by the super-run and triple-list Schur kernels only:
  // the covisibility-group kernel and the back-substitution invert in registers
  if (d.M > 0 && (d.n_sup > 0 || d.n_tchunk > 0))
    BA_LAUNCH(K_DAMP_INVERT, k_damp_invert, dim3(cdiv(d.M, kBlock)), dim3(kBlock), s, d, 0);
}
// reader variant: also stores the damped C_i (ba_get_C)
void launch_damp_invert_export(const DevProblem &d, hipStream_t s) {
  if (d.M > 0)
    hipLaunchKernelGGL(k_damp_invert, dim3(cdiv(d.M, kBlock)), dim3(kBlock), 0, s, d, 1);
}

#ifdef BA_LL_DBG
extern "C" int ba_debug_read_ll(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ll_dbg), sizeof(long long) * 32);
}
#endif
#ifdef BA_LG_DBG
extern "C" int ba_debug_read_lg(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lg_dbg), sizeof(long long) * 64);
}
#endif
#ifdef BA_GRP_DBG
extern "C" int ba_debug_read_grp(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grp_dbg), sizeof(long long) * 256);
}
#endif
#ifdef BA_SCHUR_DBG
extern "C" int ba_debug_read(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_schur_dbg), sizeof(long long) * 4 * 160);
}
#endif
void launch_schur_accumulate(const DevProblem &d, hipStream_t s) {
  if (d.n_grp32 > 0)
    BA_LAUNCH(K_SCHUR_GRP, k_schur_grp<2>, dim3(d.n_grp32), dim3(kBlock), s, d, d.grp32);
  if (d.n_grp64 > 0)
    BA_LAUNCH(K_SCHUR_GRP, k_schur_grp<4>, dim3(d.n_grp64), dim3(kBlock), s, d, d.grp64);
  if (d.n_grp128 > 0)
    BA_LAUNCH(K_SCHUR_GRP, k_schur_grp_wide, dim3(d.n_grp128), dim3(kBlock), s, d, d.grp128);
  if (d.n_sup > 0)
    BA_LAUNCH(K_SCHUR_LDS, k_schur_lds, dim3(d.n_sup), dim3(kBlock), s, d);
  if (d.n_tchunk > 0)
    BA_LAUNCH(K_SCHUR_PARTIAL, k_schur_partial, dim3(d.n_tchunk), dim3(64), s, d);
}

void launch_schur_final(const DevProblem &d, bool direct, hipStream_t s) {
  if (d.B <= 0) return;
  if (direct)
    BA_LAUNCH(K_SCHUR_FINAL, k_schur_final_direct, dim3((unsigned)d.B), dim3(kBlock), s, d);
  else
    BA_LAUNCH(K_SCHUR_FINAL, k_schur_final, dim3((unsigned)d.B), dim3(kBlock), s, d);
}

void launch_schur(const DevProblem &d, bool direct, bool with_init, hipStream_t s) {
  if (with_init)
    launch_dense_init(d.L, d.ld, d.col_x, d.zt_I, d.zt_J, d.n_zt, d.nb, &d.ctrl->done, s);
  launch_schur_accumulate(d, s);
  launch_schur_final(d, direct, s);
}

void launch_scatter(const DevProblem &d, hipStream_t s) {
  const int64_t n = d.B * 36 + 6 * (int64_t)d.N;
  if (n > 0)
    BA_LAUNCH(K_SCATTER, k_scatter, dim3(cdiv(n, kBlock)), dim3(kBlock), s, d);
}

void launch_backsub_update(const DevProblem &d, hipStream_t s, bool zero_tiles) {
  // pose workgroups, covisibility-group pieces, chunk workgroups, [factor-tile reset] (see the kernel)
  const dim3 grid(kPoseGrid + d.n_lm_part + (zero_tiles ? d.n_zt : 0));
  if (d.n_bs_grp > 0)
    BA_LAUNCH(K_BACKSUB_UPDATE, k_backsub_update<true>, grid, dim3(kBlock), s, d);
  else
    BA_LAUNCH(K_BACKSUB_UPDATE, k_backsub_update<false>, grid, dim3(kBlock), s, d);
}

// back-substitution + update + trial-point linearisation in one launch (see k_backsub_lin);
// false: the problem does not qualify (chunk pieces beside the groups)
bool can_fuse_backsub_lin(const DevProblem &d) {
  return d.n_bs_grp > 0 && d.n_lm_part == d.n_bs_grp && d.n_lin_desc == d.n_bs_grp && d.n_bchunk == d.lin_chunk0 &&
         d.bl_flag && d.pose_flag;
}
void launch_backsub_lin(const DevProblem &d, hipStream_t s, bool zero_tiles, int gen, int *bad) {
  const bool masked = d.n_lin_desc > d.n_lin_plain;
  static const int lead = getenv("BA_BL_LEAD") ? std::max(1, atoi(getenv("BA_BL_LEAD"))) : kBlLead;  // (tuning knob)
  const int P = d.n_bs_grp, K = std::min(P, lead), G = (P - K + 7) / 8;
  const dim3 grid(kPoseGrid + 2 * K + 16 * G + (zero_tiles ? d.n_zt : 0));
  if (d.n_cam <= kCamLds) {
    if (masked)
      BA_LAUNCH(K_BACKSUB_LIN, (k_backsub_lin<true, true>), grid, dim3(kBlock), s, d, gen, bad, lead);
    else
      BA_LAUNCH(K_BACKSUB_LIN, (k_backsub_lin<true, false>), grid, dim3(kBlock), s, d, gen, bad, lead);
  } else {
    if (masked)
      BA_LAUNCH(K_BACKSUB_LIN, (k_backsub_lin<false, true>), grid, dim3(kBlock), s, d, gen, bad, lead);
    else
      BA_LAUNCH(K_BACKSUB_LIN, (k_backsub_lin<false, false>), grid, dim3(kBlock), s, d, gen, bad, lead);
  }
}

void launch_scalars(const DevProblem &d, int cost_src, hipStream_t s) {
  BA_LAUNCH(K_SCALARS, k_scalars, dim3(1), dim3(kScalBlock), s, d, 1, cost_src, -1);
}

void launch_scalars_and_control(const DevProblem &d, int cost_src, hipStream_t s, int finalize_sel) {
  const int nfin = finalize_sel >= 0 ? cdiv((int64_t)d.N * 27, kScalBlock) : 0;
  BA_LAUNCH(K_SCALARS, k_scalars, dim3(1 + nfin), dim3(kScalBlock), s, d, 2, cost_src, finalize_sel);
}

void launch_control(const DevProblem &d, hipStream_t s) {
  BA_LAUNCH(K_CONTROL, k_control, dim3(1), dim3(64), s, d);
}

void launch_init_ctrl_cost(const DevProblem &d, hipStream_t s) {
  BA_LAUNCH(K_CONTROL, k_init_ctrl_cost, dim3(1), dim3(64), s, d);
}

// initial-cost scalar reduction (mode 0)
void launch_scalars_cost_only(const DevProblem &d, int cost_src, hipStream_t s) {
  BA_LAUNCH(K_SCALARS, k_scalars, dim3(1), dim3(kScalBlock), s, d, 0, cost_src, -1);
}

}  // namespace ba
