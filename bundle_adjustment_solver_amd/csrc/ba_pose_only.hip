// ba_pose_only.hip — pose-only 6-DoF Gauss-Newton (monocular and stereo), fp32,
// gfx950.
//
// Replaces PoseOnlyBundleAdjustmentSolver::Solve_Monocular_6Dof (reference
// core/pose_only_bundle_adjustment_solver.cpp:8-170 with helpers :1338-1452,
// :1147-1200, :1280-1316) and ::Solve_Stereo_6Dof (:172-399: the right camera
// sees X_r = left_to_right^-1 * X_l, contributes where its pixel is
// non-negative, has its own inlier mask; the error is normalised by
// (count_left + count_right) * 0.5f).  The whole GN loop runs inside ONE persistent
// launch of up to 64 co-resident workgroups (the problem is 10 k points ~ 200 KB:
// launch / PCIe latency dominates, not bandwidth): per iteration every thread
// linearises its points, the 21+6+1+1 sums are reduced through DPP + LDS and,
// across workgroups, through one grid barrier; thread 0 of every workgroup solves
// the 6x6 system with the pivoted LDL^T the reference gets from Eigen and
// composes the se3 exponential onto the pose held in LDS.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/ba_hip.h"
#include "ba_device.h"

namespace ba {

namespace {

constexpr int kPoThreads = 1024;
constexpr int kPoWaves = kPoThreads / 64;
constexpr int kPoMaxGroups = 64;   // workgroups of one launch (all co-resident)
constexpr int kPoPointsPerThread = 2;  // aimed at when choosing the group count
constexpr int kNred = 29;  // 21 upper H + 6 g + 1 err + 1 count of right-camera edges

// Wave-wide fp32 sum on the DPP network (xor-1, xor-2, half-mirror, mirror leave
// every lane with the total of its row of 16; the four row totals are added in
// order).  Fixed association; every lane returns the total.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_f32(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);  // row_half_mirror
  v += dpp_f32<0x140>(v);  // row_mirror
  return ((lane_f32(v, 0) + lane_f32(v, 16)) + lane_f32(v, 32)) + lane_f32(v, 48);
}

// Eigen-style pivoted LDL^T solve of a 6x6 system in fp32 (thread 0 only).
// m, d, tr, tmp point to LDS: the pivoted algorithm indexes them dynamically,
// which would otherwise put them in scratch (global) memory.
__device__ void ldlt6_solve(float *m /*36, row-major, lower used*/, float *d, int *tr,
                            float *tmp) {
#define AT(r, c) m[(r) * 6 + (c)]
  bool early = false;
  for (int k = 0; k < 6 && !early; ++k) {
    int big = k;
    float bigv = fabsf(AT(k, k));
    for (int i = k + 1; i < 6; ++i)
      if (fabsf(AT(i, i)) > bigv) {
        bigv = fabsf(AT(i, i));
        big = i;
      }
    tr[k] = big;
    if (k != big) {
      const int s = 6 - big - 1;
      for (int c = 0; c < k; ++c) {
        float t = AT(k, c); AT(k, c) = AT(big, c); AT(big, c) = t;
      }
      for (int r = 0; r < s; ++r) {
        float t = AT(big + 1 + r, k);
        AT(big + 1 + r, k) = AT(big + 1 + r, big);
        AT(big + 1 + r, big) = t;
      }
      {
        float t = AT(k, k); AT(k, k) = AT(big, big); AT(big, big) = t;
      }
      for (int i = k + 1; i < big; ++i) {
        float t = AT(i, k); AT(i, k) = AT(big, i); AT(big, i) = t;
      }
    }
    const int rs = 6 - k - 1;
    if (k > 0) {
      float acc = 0.0f;
      for (int c = 0; c < k; ++c) {
        tmp[c] = AT(c, c) * AT(k, c);
        acc += AT(k, c) * tmp[c];
      }
      AT(k, k) -= acc;
      for (int r = 0; r < rs; ++r) {
        float s2 = 0.0f;
        for (int c = 0; c < k; ++c) s2 += AT(k + 1 + r, c) * tmp[c];
        AT(k + 1 + r, k) -= s2;
      }
    }
    const float akk = AT(k, k);
    const bool valid = fabsf(akk) > 0.0f;
    if (k == 0 && !valid) {
      for (int j = 0; j < 6; ++j) tr[j] = j;
      early = true;
      break;
    }
    if (rs > 0 && valid)
      for (int r = 0; r < rs; ++r) AT(k + 1 + r, k) /= akk;
  }
  for (int i = 0; i < 6; ++i)
    if (tr[i] != i) { float t = d[i]; d[i] = d[tr[i]]; d[tr[i]] = t; }
  for (int i = 0; i < 6; ++i) {
    float s = d[i];
    for (int c = 0; c < i; ++c) s -= AT(i, c) * d[c];
    d[i] = s;
  }
  for (int i = 0; i < 6; ++i) {
    if (fabsf(AT(i, i)) > 1.17549435e-38f) d[i] /= AT(i, i);
    else d[i] = 0.0f;
  }
  for (int i = 5; i >= 0; --i) {
    float s = d[i];
    for (int r = i + 1; r < 6; ++r) s -= AT(r, i) * d[r];
    d[i] = s;
  }
  for (int i = 5; i >= 0; --i)
    if (tr[i] != i) { float t = d[i]; d[i] = d[tr[i]]; d[tr[i]] = t; }
#undef AT
}

// One camera's terms of one point (reference :1350-1452) added to acc.
__device__ __forceinline__ void po_edge(const float Lp[3], float fx, float fy, float cx,
                                        float cy, float pu, float pv, float thr_huber,
                                        float thr_out, float acc[kNred], uint8_t *mk) {
  // reference :1350-1384
  const float iz = 1.0f / Lp[2];
  const float xiz = Lp[0] * iz, yiz = Lp[1] * iz;
  const float fxxiz = fx * xiz, fyyiz = fy * yiz;
  const float ru = (fxxiz + cx) - pu;
  const float rv = (fyyiz + cy) - pv;
  float Ju[6], Jv[6];
  Ju[0] = fx * iz; Ju[1] = 0.0f; Ju[2] = -fxxiz * iz; Ju[3] = -fxxiz * yiz;
  Ju[4] = fx * (1.0f + xiz * xiz); Ju[5] = -fx * yiz;
  Jv[0] = 0.0f; Jv[1] = fy * iz; Jv[2] = -fyyiz * iz;
  Jv[3] = -fy * (1.0f + yiz * yiz); Jv[4] = fyyiz * xiz; Jv[5] = fy * xiz;
  // reference :1386-1452
  const float ars = fabsf(ru) + fabsf(rv);
  const bool hub = ars >= thr_huber;
  const float w = hub ? thr_huber / ars : 1.0f;
  const float wru = hub ? w * ru : ru, wrv = hub ? w * rv : rv;
  // per-edge Hessian first, then added to the running sums (the reference forms
  // hessian_i and appends it: same association)
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) {
      float hv = 0.0f;
      if (r != 1 && c != 1) hv += hub ? (w * Ju[r]) * Ju[c] : Ju[r] * Ju[c];
      if (r != 0 && c != 0) hv += hub ? (w * Jv[r]) * Jv[c] : Jv[r] * Jv[c];
      acc[k++] += hv;
    }
#pragma unroll
  for (int c = 0; c < 6; ++c) acc[21 + c] -= wru * Ju[c] + wrv * Jv[c];
  acc[27] += hub ? wru * ru : rv * rv;  // reference :1432,:1450 (Q9)
  if (ars >= thr_out) *mk = 0;          // reference :95-98 / :287-290, :321-324
}

// meta[0] = iterations executed, meta[1] = converged, meta[2] = rows logged,
// meta[3] = success (0 = NaN)
// STEREO: uvr2 = right pixels, cam_r = {fx,fy,cx,cy, Rrl(9), trl(3)} with
// (Rrl, trl) = left_to_right^-1, maskr = right inlier mask.
// Grid-wide barrier of the co-resident workgroups of ONE launch (gridDim.x <=
// kPoMaxGroups workgroups of 1024 threads always fit the 256 CUs at once).
// gsync[0] counts arrivals, monotonically: barrier number b is passed when it
// reaches (b+1)*gridDim.x.  The release add publishes this workgroup's partial
// sums, the acquire load makes the others' visible (and invalidates the CU's
// L1 for every wave of the workgroup).  The spin is bounded: on a time-out
// gsync[1] is set and the kernel still runs to its end, so the grid always
// drains; the host then reports an error.
__device__ __forceinline__ void po_grid_barrier(int *gsync, int target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&gsync[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(&gsync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 22)) {
        gsync[1] = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
}

// Several workgroups (one per CU) share the points of one problem: each
// linearises its share, the 29 partial sums meet in `partial` (double-buffered
// by iteration parity), ONE grid barrier per Gauss-Newton iteration, and then
// EVERY workgroup adds the partials in the same order and runs the same 6x6
// solve, so all hold the same pose and take the same convergence decision
// without a second barrier or a broadcast.  Workgroup 0 writes the outputs.
template <bool STEREO>
__global__ __launch_bounds__(kPoThreads) void k_pose_only6(
    const float *__restrict__ X3, const float *__restrict__ uv2,
    const float *__restrict__ uvr2, int n, float fx, float fy, float cx, float cy,
    const float *__restrict__ cam_r, float *T12, uint8_t *mask, uint8_t *maskr,
    float thr_huber, float thr_step, float thr_cost, float thr_out, int max_it,
    PoIter *iters, int cap, int *meta, float *debug_T12, int *gsync, float *partial) {
  const int G = gridDim.x;
  const bool lead = blockIdx.x == 0;
  __shared__ float red[kPoWaves][kNred];
  __shared__ float tots[kNred];
  __shared__ float Hs[36], gs[6], tmps[6];
  __shared__ int trs[6];
  __shared__ float pose[12];  // camera_to_world_optimized: R (9) then t (3)
  __shared__ int ctl[2];      // [0] = stop flag
  __shared__ float s_err_prev;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    // pose = reference_to_current.inverse()  (reference :52-53)
    float R[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) R[r * 3 + c] = T12[c * 3 + r];
    for (int k = 0; k < 9; ++k) pose[k] = R[k];
    for (int r = 0; r < 3; ++r)
      pose[9 + r] = -(R[r * 3 + 0] * T12[9] + R[r * 3 + 1] * T12[10] +
                      R[r * 3 + 2] * T12[11]);
    ctl[0] = 0;
    ctl[1] = 0;  // iterations executed
    s_err_prev = 1e10f;
    if (lead) {
      meta[0] = 0;
      meta[1] = 1;
      meta[2] = 0;
      meta[3] = 1;
    }
  }
  int n_rows = 0;  // Summary rows logged so far (thread 0)
  __syncthreads();
  const float inv_n = 1.0f / (float)n;
  for (int it = 0; it < max_it; ++it) {
    float Rl[9], tl[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rl[k] = pose[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tl[k] = pose[9 + k];
    float acc[kNred];
#pragma unroll
    for (int k = 0; k < kNred; ++k) acc[k] = 0.0f;
    float cr[16];
    if (STEREO) {
#pragma unroll
      for (int k = 0; k < 16; ++k) cr[k] = cam_r[k];
    }
    // four points per trip: their loads are issued together, so a trip pays one
    // memory latency instead of four (the inputs are re-read from L2 in every
    // Gauss-Newton iteration)
    const int gstride = G * kPoThreads;
    for (int p0 = blockIdx.x * kPoThreads + tid; p0 < n; p0 += 4 * gstride) {
      float Xb[4][3], ub[4][2], urb[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + u * gstride;
        const int pc = p < n ? p : n - 1;
        Xb[u][0] = X3[3 * pc]; Xb[u][1] = X3[3 * pc + 1]; Xb[u][2] = X3[3 * pc + 2];
        ub[u][0] = uv2[2 * pc]; ub[u][1] = uv2[2 * pc + 1];
        if (STEREO) { urb[u][0] = uvr2[2 * pc]; urb[u][1] = uvr2[2 * pc + 1]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + u * gstride;
        if (p >= n) break;
        float Lp[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
          Lp[r] = (Rl[r * 3 + 0] * Xb[u][0] + Rl[r * 3 + 1] * Xb[u][1] + Rl[r * 3 + 2] * Xb[u][2]) + tl[r];
        po_edge(Lp, fx, fy, cx, cy, ub[u][0], ub[u][1], thr_huber, thr_out, acc, mask + p);
        if (STEREO) {
          const float pu = urb[u][0], pv = urb[u][1];
          if (!(pu < 0 || pv < 0)) {  // reference :298
            float Lr[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
              Lr[r] = (cr[4 + r * 3 + 0] * Lp[0] + cr[4 + r * 3 + 1] * Lp[1] +
                       cr[4 + r * 3 + 2] * Lp[2]) +
                      cr[13 + r];
            po_edge(Lr, cr[0], cr[1], cr[2], cr[3], pu, pv, thr_huber, thr_out, acc, maskr + p);
            acc[28] += 1.0f;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kNred; ++k) {
      const float s = wave_sum_f(acc[k]);
      if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    // cross-wave sums in parallel (thread k adds the 16 wave totals of sum k in
    // order), then thread 0 runs the 6x6 solve
    if (tid < kNred) {
      float sk = 0.0f;
#pragma unroll
      for (int w = 0; w < kPoWaves; ++w) sk += red[w][tid];
      tots[tid] = sk;
    }
    if (G > 1) {
      float *pb = partial + (size_t)(it & 1) * kPoMaxGroups * kNred;
      // agent-scope accesses: the partials cross XCDs (separate L2s)
      if (tid < kNred)
        __hip_atomic_store(&pb[blockIdx.x * kNred + tid], tots[tid], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      po_grid_barrier(gsync, (it + 1) * G);
      if (tid < kNred) {  // every workgroup: the same sum in the same order
        float sk = 0.0f;
        for (int w = 0; w < G; ++w)
          sk += __hip_atomic_load(&pb[w * kNred + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tots[tid] = sk;
      }
    }
    __syncthreads();
    if (tid == 0) {
      float tot[kNred];
#pragma unroll
      for (int k = 0; k < kNred; ++k) tot[k] = tots[k];
      float *H = Hs, *g = gs;
      int k = 0;
      for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) {
          H[r * 6 + c] = tot[k];
          H[c * 6 + r] = tot[k];
          ++k;
        }
      for (int r = 0; r < 6; ++r) H[r * 6 + r] *= (1.0f + 1e-5f);  // :103
      for (int c = 0; c < 6; ++c) g[c] = tot[21 + c];
      ldlt6_solve(H, g, trs, tmps);  // delta_xi, reference :105
      // se3 exponential, reference :1280-1316
      const float v0 = g[0], v1 = g[1], v2 = g[2], w0 = g[3], w1 = g[4], w2 = g[5];
      const float theta = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
      const float wx[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
      float wx2[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          wx2[r * 3 + c] = wx[r * 3 + 0] * wx[0 * 3 + c] + wx[r * 3 + 1] * wx[1 * 3 + c] +
                           wx[r * 3 + 2] * wx[2 * 3 + c];
      float ca, cb, va, vb;
      if (theta < 1e-7f) {
        ca = 1.0f; cb = 0.5f; va = 0.5f; vb = 0.33333333333333333333333333f;
      } else {
        const float st = sinf(theta), ct = cosf(theta);
        ca = st / theta;
        cb = (1.0f - ct) / (theta * theta);
        va = cb;
        vb = (theta - st) / (theta * theta * theta);
      }
      float dR[9], V[9], dt[3];
      for (int q = 0; q < 9; ++q) {
        const float id = (q % 4 == 0) ? 1.0f : 0.0f;
        dR[q] = id + ca * wx[q] + cb * wx2[q];
        V[q] = id + va * wx[q] + vb * wx2[q];
      }
      for (int r = 0; r < 3; ++r) dt[r] = V[r * 3 + 0] * v0 + V[r * 3 + 1] * v1 + V[r * 3 + 2] * v2;
      float Rn[9], tn[3];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          Rn[r * 3 + c] = dR[r * 3 + 0] * pose[0 * 3 + c] + dR[r * 3 + 1] * pose[1 * 3 + c] +
                          dR[r * 3 + 2] * pose[2 * 3 + c];
        tn[r] = dR[r * 3 + 0] * pose[9] + dR[r * 3 + 1] * pose[10] + dR[r * 3 + 2] * pose[11] + dt[r];
      }
      for (int q = 0; q < 9; ++q) pose[q] = Rn[q];
      for (int q = 0; q < 3; ++q) pose[9 + q] = tn[q];
      if (lead && debug_T12 && it < cap) {
        float *D = debug_T12 + 12 * it;
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) D[r * 3 + c] = Rn[c * 3 + r];
        for (int r = 0; r < 3; ++r)
          D[9 + r] = -(D[r * 3 + 0] * tn[0] + D[r * 3 + 1] * tn[1] + D[r * 3 + 2] * tn[2]);
      }
      // mono: reference :112; stereo: :331 (count_left = n, exact in fp32 below 2^24)
      const float err_curr = STEREO ? tot[27] / (((float)n + tot[28]) * 0.5f)
                                    : tot[27] * (inv_n * 0.5f);
      const float delta_error = fabsf(err_curr - s_err_prev);
      const float dn = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + w0 * w0 + w1 * w1 + w2 * w2);
      ctl[1] = it + 1;
      if (lead) meta[0] = it + 1;
      if (dn < thr_step || delta_error < thr_cost) {
        if (lead) meta[1] = 1;
        ctl[0] = 1;  // converged: no Summary row (reference :116-121)
      } else {
        if (lead && it == max_it - 1) meta[1] = 0;
        if (lead && iters && n_rows < cap) {
          iters[n_rows].cost = err_curr;
          iters[n_rows].cost_change = delta_error;
          iters[n_rows].abs_step = dn;
        }
        ++n_rows;
        if (lead) meta[2] = n_rows;
        s_err_prev = err_curr;
      }
    }
    __syncthreads();
    if (ctl[0]) break;
  }
  // (with no iteration executed nothing was exchanged: T12 stays as given, and
  //  no workgroup may write it while another still reads it)
  if (tid == 0 && lead && ctl[1] > 0) {
    float nrm = 0.0f;
    for (int k = 0; k < 9; ++k) nrm += pose[k] * pose[k];
    if (isnan(nrm)) {
      meta[3] = 0;  // reference :159-167: do not update on NaN
    } else {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T12[r * 3 + c] = pose[c * 3 + r];
      for (int r = 0; r < 3; ++r)
        T12[9 + r] = -(T12[r * 3 + 0] * pose[9] + T12[r * 3 + 1] * pose[10] +
                       T12[r * 3 + 2] * pose[11]);
    }
  }
}

}  // namespace

// workgroups for n points: about kPoPointsPerThread points per thread
static int po_groups(int n) {
  int g = (n + kPoThreads * kPoPointsPerThread - 1) / (kPoThreads * kPoPointsPerThread);
  return g < 1 ? 1 : (g > kPoMaxGroups ? kPoMaxGroups : g);
}
int pose_only_sync_ints() { return 2; }
int pose_only_partial_floats() { return 2 * kPoMaxGroups * kNred; }

int pose_only_mono6_device(const float *dX3, const float *duv2, int n, float fx,
                           float fy, float cx, float cy, float *dT12,
                           uint8_t *dmask, float thr_huber, float thr_step,
                           float thr_cost, float thr_out, int max_it,
                           PoIter *d_iters, int cap, int *d_meta,
                           float *d_debug, int *d_gsync, float *d_partial, hipStream_t s) {
  // d_gsync must be zero on entry (the caller's single H2D copy covers it)
  hipLaunchKernelGGL(k_pose_only6<false>, dim3(po_groups(n)), dim3(kPoThreads), 0, s, dX3, duv2,
                     (const float *)nullptr, n, fx, fy, cx, cy, (const float *)nullptr, dT12,
                     dmask, (uint8_t *)nullptr, thr_huber, thr_step, thr_cost, thr_out, max_it,
                     d_iters, cap, d_meta, d_debug, d_gsync, d_partial);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int pose_only_stereo6_device(const float *dX3, const float *duvl2, const float *duvr2, int n,
                             float fx, float fy, float cx, float cy, const float *d_cam_r16,
                             float *dT12, uint8_t *dmask_l, uint8_t *dmask_r, float thr_huber,
                             float thr_step, float thr_cost, float thr_out, int max_it,
                             PoIter *d_iters, int cap, int *d_meta, float *d_debug,
                             int *d_gsync, float *d_partial, hipStream_t s) {
  // d_gsync must be zero on entry (the caller's single H2D copy covers it)
  hipLaunchKernelGGL(k_pose_only6<true>, dim3(po_groups(n)), dim3(kPoThreads), 0, s, dX3, duvl2,
                     duvr2, n, fx, fy, cx, cy, d_cam_r16, dT12, dmask_l, dmask_r, thr_huber,
                     thr_step, thr_cost, thr_out, max_it, d_iters, cap, d_meta, d_debug, d_gsync,
                     d_partial);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace ba
