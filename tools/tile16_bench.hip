// Micro-benchmark of the 16x16 tile factorisation (csrc/ba_tile16.h): cycles per
// call of one wave (s_memtime), phase stamps, and accuracy against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -I bundle_adjustment_solver_amd/csrc tools/tile16_bench.hip -o tools/tile16_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ long long g_stamp[32];
__device__ int g_stamp_on;
#define BA_T16_STAMP(k)                                                    \
  {                                                                        \
    if (g_stamp_on) {                                                      \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          \
      if (threadIdx.x == 0) g_stamp[(k)] = __builtin_readcyclecounter();   \
    }                                                                      \
  }
#include "ba_tile16.h"

using namespace ba::tile16;

template <int VARIANT>
__global__ __launch_bounds__(64) void k_bench(const double *A, double *out, long long *cyc, int reps, int stamp) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  double g0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * j + q;
    g0[j] = (r >= c) ? A[r * 16 + c] : 0.0;
  }
  g_stamp_on = 0;
  double g[4], dinv = 0.0, sink = 0.0;
  int nbad = 0;
  long long best = 1ll << 60;
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = g0[j] + sink * 1e-300;  // (dependence on the previous call)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t0 = __builtin_readcyclecounter();
    if (VARIANT == 0) nbad = tile16_potrf_inv(g, lane, dinv);
    else nbad = tile16_potrf_inv2(g, lane, dinv);
    sink = g[0] + g[1] + g[2] + g[3] + dinv;
    asm volatile("" ::"v"(sink));
    const long long t1 = __builtin_readcyclecounter();
    if (t1 - t0 < best) best = t1 - t0;
  }
  if (stamp) {
    g_stamp_on = 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = g0[j];
    if (VARIANT == 0) nbad = tile16_potrf_inv(g, lane, dinv);
    else nbad = tile16_potrf_inv2(g, lane, dinv);
    g_stamp_on = 0;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) out[(4 * j + q) + 16 * r] = g[j];  // out[r][c]
  if (r == (4 * 0 + q) || true) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (r == 4 * j + q) out[256 + r] = dinv;
  }
  if (lane == 0) {
    cyc[0] = best;
    cyc[1] = nbad;
  }
}

// ---- the 32x32 one-wave routine ----
__global__ __launch_bounds__(64) void k_bench32(const double *A, double *out, long long *cyc, int reps) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  double a00[4], a10[4], a11[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * j + q;
    a00[j] = (r >= c) ? A[r * 32 + c] : 0.0;
    a10[j] = A[(16 + r) * 32 + c];
    a11[j] = (r >= c) ? A[(16 + r) * 32 + 16 + c] : 0.0;
  }
  g_stamp_on = 0;
  double g00[4], g10[4], g11[4], d0 = 0.0, d1 = 0.0, sink = 0.0;
  int bad = 0;
  long long best = 1ll << 60;
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g00[j] = a00[j] + sink * 1e-300;
      g10[j] = a10[j];
      g11[j] = a11[j];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t0 = __builtin_readcyclecounter();
    bad = tile32_potrf_inv(g00, g10, g11, lane, d0, d1);
    sink = g00[0] + g10[1] + g11[2] + d0 + d1;
    asm volatile("" ::"v"(sink));
    const long long t1 = __builtin_readcyclecounter();
    if (t1 - t0 < best) best = t1 - t0;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * j + q;
    out[r * 32 + c] = g00[j];
    out[(16 + r) * 32 + c] = g10[j];
    out[(16 + r) * 32 + 16 + c] = g11[j];
    if (r == c) {
      out[1024 + r] = d0;
      out[1024 + 16 + r] = d1;
    }
  }
  if (lane == 0) {
    cyc[0] = best;
    cyc[1] = bad;
  }
}

static int run32(double *dA, double *dout, long long *dcyc) {
  // SPD 32x32
  std::vector<double> A(1024), Q(32 * 40);
  for (double &v : Q) v = rand() / (double)RAND_MAX - 0.5;
  for (int r = 0; r < 32; ++r)
    for (int c = 0; c < 32; ++c) {
      double s = (r == c) ? 0.3 : 0.0;
      for (int k = 0; k < 40; ++k) s += Q[r * 40 + k] * Q[c * 40 + k];
      A[r * 32 + c] = s * 1e4;
    }
  std::vector<double> L(1024, 0.0);
  for (int c = 0; c < 32; ++c) {
    double d = A[c * 32 + c];
    for (int k = 0; k < c; ++k) d -= L[c * 32 + k] * L[c * 32 + k];
    L[c * 32 + c] = std::sqrt(d);
    for (int r2 = c + 1; r2 < 32; ++r2) {
      double v = A[r2 * 32 + c];
      for (int k = 0; k < c; ++k) v -= L[r2 * 32 + k] * L[c * 32 + k];
      L[r2 * 32 + c] = v / L[c * 32 + c];
    }
  }
  // E_tt = inverse transpose of the diagonal 16x16 blocks of L
  std::vector<double> E(2 * 256, 0.0);
  for (int t = 0; t < 2; ++t) {
    std::vector<double> Li(256, 0.0);
    auto Lt = [&](int i, int j) { return L[(16 * t + i) * 32 + 16 * t + j]; };
    for (int c = 0; c < 16; ++c) {
      Li[c * 16 + c] = 1.0 / Lt(c, c);
      for (int r2 = c + 1; r2 < 16; ++r2) {
        double v = 0.0;
        for (int k = c; k < r2; ++k) v -= Lt(r2, k) * Li[k * 16 + c];
        Li[r2 * 16 + c] = v / Lt(r2, r2);
      }
    }
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) E[t * 256 + i * 16 + j] = Li[j * 16 + i];
  }
  double *dA32, *dout32;
  hipMalloc(&dA32, 1024 * 8);
  hipMalloc(&dout32, (1024 + 32) * 8);
  hipMemset(dout32, 0, (1024 + 32) * 8);
  hipMemcpy(dA32, A.data(), 1024 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_bench32, dim3(1), dim3(64), 0, 0, dA32, dout32, dcyc, 200);
  if (hipDeviceSynchronize() != hipSuccess) { printf("tile32: kernel failed\n"); return 1; }
  std::vector<double> out(1024 + 32);
  long long cyc[2];
  hipMemcpy(out.data(), dout32, out.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(cyc, dcyc, sizeof(cyc), hipMemcpyDeviceToHost);
  double eL = 0, nL = 0, eE = 0, nE = 0, eD = 0;
  for (int r2 = 0; r2 < 32; ++r2)
    for (int c = 0; c < 32; ++c) {
      const int tr = r2 / 16, tc = c / 16;
      if (r2 >= c) {
        eL = fmax(eL, fabs(out[r2 * 32 + c] - L[r2 * 32 + c]));
        nL = fmax(nL, fabs(L[r2 * 32 + c]));
      } else if (tr == tc) {  // strict upper of a diagonal tile: its L^-T
        const double ref = E[tr * 256 + (r2 % 16) * 16 + (c % 16)];
        eE = fmax(eE, fabs(out[r2 * 32 + c] - ref));
        nE = fmax(nE, fabs(ref));
      }
    }
  for (int i = 0; i < 32; ++i) eD = fmax(eD, fabs(out[1024 + i] - E[(i / 16) * 256 + (i % 16) * 17]) / fabs(E[(i / 16) * 256 + (i % 16) * 17]));
  printf("tile32     %6lld cycles/call  bad %lld  rel err L %.2e  L^-T (diagonal tiles) %.2e  diag %.2e\n", cyc[0], cyc[1],
         eL / nL, eE / nE, eD);
  return (eL > 1e-12 * nL || eE > 1e-10 * nE || cyc[1] != 0) ? 1 : 0;
}

static void host_ref(const std::vector<double> &A, std::vector<double> &L, std::vector<double> &E) {
  // Cholesky L (lower) and E = L^-T (upper)
  L.assign(256, 0.0);
  for (int c = 0; c < 16; ++c) {
    double d = A[c * 16 + c];
    for (int k = 0; k < c; ++k) d -= L[c * 16 + k] * L[c * 16 + k];
    L[c * 16 + c] = std::sqrt(d);
    for (int r2 = c + 1; r2 < 16; ++r2) {
      double v = A[r2 * 16 + c];
      for (int k = 0; k < c; ++k) v -= L[r2 * 16 + k] * L[c * 16 + k];
      L[r2 * 16 + c] = v / L[c * 16 + c];
    }
  }
  // Linv (lower), then E = Linv^T
  std::vector<double> Li(256, 0.0);
  for (int c = 0; c < 16; ++c) {
    Li[c * 16 + c] = 1.0 / L[c * 16 + c];
    for (int r2 = c + 1; r2 < 16; ++r2) {
      double v = 0.0;
      for (int k = c; k < r2; ++k) v -= L[r2 * 16 + k] * Li[k * 16 + c];
      Li[r2 * 16 + c] = v / L[r2 * 16 + r2];
    }
  }
  E.assign(256, 0.0);
  for (int r2 = 0; r2 < 16; ++r2)
    for (int c = 0; c < 16; ++c) E[r2 * 16 + c] = Li[c * 16 + r2];
}

template <int V>
static int run(const char *name, const std::vector<double> &A, double *dA, double *dout, long long *dcyc, bool zero_row) {
  std::vector<double> L, E;
  host_ref(A, L, E);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_bench<V>, dim3(1), dim3(64), 0, 0, dA, dout, dcyc, 200, 1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel failed\n", name); return 1; }
  std::vector<double> out(256 + 16);
  long long cyc[2];
  hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(cyc, dcyc, sizeof(cyc), hipMemcpyDeviceToHost);
  long long st[32];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
  double eL = 0, eE = 0, eD = 0, nL = 0, nE = 0;
  if (!zero_row)
    for (int r2 = 0; r2 < 16; ++r2) {
      for (int c = 0; c < 16; ++c) {
        if (r2 >= c) { eL = fmax(eL, fabs(out[r2 * 16 + c] - L[r2 * 16 + c])); nL = fmax(nL, fabs(L[r2 * 16 + c])); }
        else { eE = fmax(eE, fabs(out[r2 * 16 + c] - E[r2 * 16 + c])); nE = fmax(nE, fabs(E[r2 * 16 + c])); }
      }
      eD = fmax(eD, fabs(out[256 + r2] - E[r2 * 16 + r2]) / fabs(E[r2 * 16 + r2]));
    }
  printf("%-10s %6lld cycles/call  bad pivots %lld  rel err L %.2e  L^-T %.2e  diag(L^-T) %.2e\n", name, cyc[0], cyc[1],
         nL > 0 ? eL / nL : 0.0, nE > 0 ? eE / nE : 0.0, eD);
  printf("           phase cycles per block [fetch, pivots, y, select, mfma]:");
  for (int kb = 0; kb < 4; ++kb) {
    printf("  |");
    for (int p = 0; p < 4; ++p) printf(" %lld", st[5 * kb + p + 1] - st[5 * kb + p]);
  }
  printf("  | final %lld  (stamped total %lld)\n", st[20] - st[19], st[20] - st[0]);
  if (zero_row) {
    printf("           zero pivot case: row/col 5 of L: ");
    for (int c = 0; c < 16; ++c) printf("%.2g ", out[5 * 16 + c]);
    printf(" dinv %.3g\n", out[256 + 5]);
  }
  return (!zero_row && (eL > 1e-12 * nL || eE > 1e-10 * nE)) ? 1 : 0;
}

int main() {
  std::vector<double> A(256);
  srand(7);
  std::vector<double> Q(16 * 24);
  for (double &v : Q) v = rand() / (double)RAND_MAX - 0.5;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      double s = (r == c) ? 0.3 : 0.0;
      for (int k = 0; k < 24; ++k) s += Q[r * 24 + k] * Q[c * 24 + k];
      A[r * 16 + c] = s * 1e4;
    }
  double *dA, *dout;
  long long *dcyc;
  hipMalloc(&dA, 256 * 8);
  hipMalloc(&dout, (256 + 16) * 8);
  hipMalloc(&dcyc, 16);
  int fail = 0;
  fail += run<0>("current", A, dA, dout, dcyc, false);
  fail += run<1>("new", A, dA, dout, dcyc, false);
  // a pose without observations: zero row and column 5
  std::vector<double> Z(A);
  for (int k = 0; k < 16; ++k) Z[5 * 16 + k] = Z[k * 16 + 5] = 0.0;
  run<0>("current/0", Z, dA, dout, dcyc, true);
  run<1>("new/0", Z, dA, dout, dcyc, true);
  fail += run32(dA, dout, dcyc);
  printf(fail ? "TILE16 BENCH FAILED\n" : "TILE16 BENCH OK\n");
  return fail;
}
