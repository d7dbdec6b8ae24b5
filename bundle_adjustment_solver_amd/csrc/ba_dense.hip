// ba_dense.hip — dense solve of the reduced camera system on gfx950.
//
// Replaces reference core/full_bundle_adjustment_solver.cpp:890-908
// (`x = Am_BCinvBt_mat.ldlt().solve(rhs)`, Eigen's unblocked pivoted LDLT)
// with a blocked right-looking Cholesky whose trailing update runs on the
// fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Storage: column-major lower triangle, `ld` rows, `npad` columns (npad is a
// multiple of 64; padded diagonal = 1).  The right-hand side rides along as
// matrix ROW `npad`, so the forward substitution L z = rhs is performed by the
// panel TRSM / trailing update for free; only L^T x = z needs its own sweep.
// A non-positive pivot (pose without observations -> zero row/column) is
// treated like Eigen's pseudo-inverted D entry: the column and the solution
// component are set to zero.
#include "ba_device.h"

namespace ba {

namespace {

constexpr int NB = kDenseNb;  // 64
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_dense_init(double *L, int npad, int ld,
                                                    int n_valid,
                                                    const int *done) {
  if (done && *done) return;
  for (int c = blockIdx.x; c < npad; c += gridDim.x) {
    double *col = L + (size_t)c * ld;
    const bool pad = c >= n_valid;
    for (int r = threadIdx.x; r < ld; r += blockDim.x)
      col[r] = (pad && r == c) ? 1.0 : 0.0;
  }
}

// ---- step 1: Cholesky of the 64x64 diagonal block (one workgroup) ---------
// Blocked by 16 inside LDS.  Writes the factor to Ldiag[kb] (column-major
// 64x64, strictly-upper part zero) and 1/diag to dinv.
__global__ __launch_bounds__(256) void k_chol_diag(double *L, int ld, int k0,
                                                   double *Ldiag_k,
                                                   double *dinv_k,
                                                   const int *done) {
  if (done && *done) return;
  constexpr int LS = NB + 1;
  __shared__ double A[NB * LS];  // A[c*LS + r]
  __shared__ double inv[NB];
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int c = e / NB, r = e % NB;
    A[c * LS + r] = (r >= c) ? L[(size_t)(k0 + c) * ld + k0 + r] : 0.0;
  }
  __syncthreads();
  for (int c0 = 0; c0 < NB; c0 += 16) {
    // (i) 16x16 diagonal sub-block, lanes 0..15 of wave 0, right-looking
    if (tid < 64) {
      for (int c = 0; c < 16; ++c) {
        const double dcc = A[(c0 + c) * LS + c0 + c];
        const bool ok = dcc > 1e-300;
        const double s = ok ? sqrt(dcc) : 0.0;
        const double iv = ok ? 1.0 / s : 0.0;
        double l = 0.0;
        if (tid > c && tid < 16) {
          l = A[(c0 + c) * LS + c0 + tid] * iv;
          A[(c0 + c) * LS + c0 + tid] = l;
        }
        if (tid == c) {
          A[(c0 + c) * LS + c0 + c] = s;
          inv[c0 + c] = iv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid > c && tid < 16)
          for (int c2 = c + 1; c2 <= tid; ++c2)
            A[(c0 + c2) * LS + c0 + tid] -= l * A[(c0 + c) * LS + c0 + c2];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();
    const int rem = NB - c0 - 16;  // rows below the sub-block
    // (ii) TRSM of the rows below: one thread per row
    if (tid < rem) {
      const int r = c0 + 16 + tid;
      double xr[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double s = A[(c0 + c) * LS + r];
#pragma unroll
        for (int k = 0; k < c; ++k) s -= xr[k] * A[(c0 + k) * LS + c0 + c];
        xr[c] = s * inv[c0 + c];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) A[(c0 + c) * LS + r] = xr[c];
    }
    __syncthreads();
    // (iii) trailing update inside the block (lower part)
    for (int e = tid; e < rem * rem; e += 256) {
      const int rr = e % rem, cc = e / rem;
      if (rr < cc) continue;
      const int r = c0 + 16 + rr, c = c0 + 16 + cc;
      double s = A[c * LS + r];
#pragma unroll
      for (int k = 0; k < 16; ++k)
        s -= A[(c0 + k) * LS + r] * A[(c0 + k) * LS + c];
      A[c * LS + r] = s;
    }
    __syncthreads();
  }
  for (int e = tid; e < NB * NB; e += 256) {
    const int c = e / NB, r = e % NB;
    Ldiag_k[e] = (r >= c) ? A[c * LS + r] : 0.0;
  }
  if (tid < NB) dinv_k[tid] = inv[tid];
}

// ---- step 2: TRSM of the rows below the diagonal block --------------------
// X = A21 * L11^-T, one thread per row, the row kept in registers; L11 is
// wave-uniform and read through the scalar path.
__global__ __launch_bounds__(256) void k_chol_trsm(double *L, int ld, int k0,
                                                   int n_rows_total,
                                                   const double *__restrict__ Ld,
                                                   const double *__restrict__ dinv,
                                                   const int *done) {
  if (done && *done) return;
  const int r = k0 + NB + blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows_total) return;
  double x[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) x[c] = L[(size_t)(k0 + c) * ld + r];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    double s = x[c];
#pragma unroll
    for (int k = 0; k < c; ++k) s -= x[k] * Ld[k * NB + c];
    x[c] = s * dinv[c];
  }
#pragma unroll
  for (int c = 0; c < NB; ++c) L[(size_t)(k0 + c) * ld + r] = x[c];
}

// ---- step 3: trailing update C_IJ -= P_I P_J^T on the fp64 matrix cores ---
// 64x64 tile per workgroup, 4 waves of 32x32 (2x2 MFMA 16x16x4 tiles).
// MFMA orientation: the MFMA "row" index runs over C's COLUMN j and the MFMA
// "column" index (lane&15) over C's ROW i, so that each accumulator register
// is 16 consecutive rows of one column = 128 contiguous bytes in memory.
__global__ __launch_bounds__(256) void k_chol_syrk(double *L, int ld, int k0,
                                                   int kb, int ncb,
                                                   const int *done) {
  if (done && *done) return;
  const int Ip = blockIdx.x, Jp = blockIdx.y;  // tile offsets past block kb
  if (Jp > Ip) return;
  const int I = kb + 1 + Ip, J = kb + 1 + Jp;
  if (J >= ncb) return;  // the rhs row block has no diagonal tile
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wi = wv & 1, wj = wv >> 1;
  const int i0 = I * NB + 32 * wi, j0 = J * NB + 32 * wj;
  const int lr = lane & 15, lk = lane >> 4;
  v4f64 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};
  const double *P = L + (size_t)k0 * ld;
#pragma unroll 4
  for (int kk = 0; kk < NB / 4; ++kk) {
    const double *col = P + (size_t)(kk * 4 + lk) * ld;
    const double a0 = col[j0 + lr], a1 = col[j0 + 16 + lr];
    const double b0 = col[i0 + lr], b1 = col[i0 + 16 + lr];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = j0 + 16 * m + lk + 4 * g;
        const int i = i0 + 16 * n + lr;
        double *dst = L + (size_t)j * ld + i;
        *dst -= acc[m][n][g];
      }
}

// ---- backward sweep L^T x = z, one launch per column block (right-looking) -
// Every workgroup first solves the 64x64 diagonal system for x_k (wave 0,
// redundantly); workgroup 0 stores x_k; workgroup g>0 subtracts the strip
// L[k-block rows, column block g-1]^T x_k from z.
__global__ __launch_bounds__(256) void k_chol_back(double *L, int ld, int npad,
                                                   int kb,
                                                   const double *__restrict__ Ld,
                                                   const double *__restrict__ dinv,
                                                   double *x, int n_x,
                                                   const int *done) {
  if (done && *done) return;
  __shared__ double xs[NB];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x;
  const int k0 = kb * NB;
  if (tid < 64) {
    double w = L[(size_t)(k0 + tid) * ld + npad];
    for (int r = NB - 1; r >= 0; --r) {
      const double xr = __shfl(w * dinv[r], r, 64);
      if (tid < r) w -= Ld[tid * NB + r] * xr;
      if (tid == r) w = xr;
    }
    xs[tid] = w;
    if (blockIdx.x == 0 && k0 + tid < n_x) x[k0 + tid] = w;
  }
  if (blockIdx.x == 0) return;
  __syncthreads();
  const int cb = blockIdx.x - 1;
  const int c = cb * NB + (tid & 63), q = tid >> 6;
  const double *col = L + (size_t)c * ld + k0 + 16 * q;
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += col[r] * xs[16 * q + r];
  part[q][tid & 63] = s;
  __syncthreads();
  if (tid < 64) {
    const double tot = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    L[(size_t)c * ld + npad] -= tot;
  }
}

}  // namespace

void launch_dense_init(double *L, int npad, int ld, int n_valid,
                       const int *done_flag, hipStream_t s) {
  hipLaunchKernelGGL(k_dense_init, dim3(2048), dim3(256), 0, s, L, npad, ld,
                     n_valid, done_flag);
}

// Ldiag: (npad/64) blocks of 64*64 doubles followed by npad doubles of 1/diag.
void dense_factor_solve(double *L, int npad, int ld, double *Ldiag, double *x,
                        int n_x, const int *done, hipStream_t s) {
  const int ncb = npad / NB;
  double *dinv = Ldiag + (size_t)ncb * NB * NB;
  const int n_rows_total = npad + 1;  // rows that carry data (rhs = row npad)
  for (int kb = 0; kb < ncb; ++kb) {
    const int k0 = kb * NB;
    double *Ld = Ldiag + (size_t)kb * NB * NB;
    hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, s, L, ld, k0, Ld,
                       dinv + k0, done);
    const int rows_below = n_rows_total - (k0 + NB);
    if (rows_below > 0)
      hipLaunchKernelGGL(k_chol_trsm, dim3((rows_below + 255) / 256), dim3(256),
                         0, s, L, ld, k0, n_rows_total, Ld, dinv + k0, done);
    const int T = ncb - 1 - kb;  // remaining column blocks
    if (T > 0)
      hipLaunchKernelGGL(k_chol_syrk, dim3(T + 1, T), dim3(256), 0, s, L, ld,
                         k0, kb, ncb, done);
  }
  for (int kb = ncb - 1; kb >= 0; --kb) {
    double *Ld = Ldiag + (size_t)kb * NB * NB;
    hipLaunchKernelGGL(k_chol_back, dim3(1 + kb), dim3(256), 0, s, L, ld, npad,
                       kb, Ld, dinv + kb * NB, x, n_x, done);
  }
}

void launch_dense_solve(const DevProblem &d, hipStream_t s) {
  dense_factor_solve(d.L, d.npad, d.ld, d.Ldiag, d.x, 6 * d.N, &d.ctrl->done,
                     s);
}

}  // namespace ba
