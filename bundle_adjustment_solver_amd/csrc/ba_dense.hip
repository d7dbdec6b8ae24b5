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
#include "ba_dense_sched.h"

#include <cstdlib>
#include <vector>

namespace ba {

namespace {

constexpr int NB = kDenseNb;   // tile order (32 or 64)
constexpr int NP = NB / 16;    // 16-column panels per tile
static_assert(NB == 32 || NB == 64, "tile order");
static_assert(kDenseWsPerBlock == NB * NB + NP * 256, "workspace layout");
typedef double v4f64 __attribute__((ext_vector_type(4)));

// (Re)initialise the tiles of L that the factorisation touches: the
// structurally non-zero tiles of the factor (incl. fill-in), the diagonal tiles
// (unit diagonal on padding columns) and the rhs row block.  Every other tile
// was zeroed once at ba_finalize and is never written.
__global__ __launch_bounds__(256) void k_dense_init(double *L, int ld,
                                                    const int *__restrict__ col_x,
                                                    const int *__restrict__ zt_I,
                                                    const int *__restrict__ zt_J,
                                                    const int *done) {
  if (done && *done) return;
  const int I = zt_I[blockIdx.x], J = zt_J[blockIdx.x];
  for (int e = threadIdx.x; e < NB * NB; e += 256) {
    const int c = J * NB + e / NB, r = I * NB + e % NB;
    L[(size_t)c * ld + r] = (r == c && col_x[c] < 0) ? 1.0 : 0.0;
  }
}

// ---- step 1: Cholesky of the 64x64 diagonal block (one workgroup) ---------
// Left-looking over four 16-column panels held in LDS.  The bulk (panel
// update, TRSM of the rows below) runs on the fp64 matrix cores; the only
// serial part is the register-resident 16x16 tile factorisation below, which
// produces L_T and E_T = L_T^-T in the same 16 steps (the column operations
// that turn A into L turn I into L^-T).
//
// Outputs per block (workspace `ws`, kDenseWsPerBlock doubles):
//   ws[0 .. 4095]      L11, column-major 64x64, zero above the diagonal
//   ws[4096 + 256 p..] E_pp = L_pp^-T (16x16, row-major, upper incl. diag)
__device__ __forceinline__ double readlane_f64(double v, int src) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}

// Lane (r = lane&15, q = lane>>4) holds g[j] = G[r][4j+q].  On entry the lower
// triangle (r >= c) is the SPD tile and the strict upper part is 0; on exit the
// lower triangle is L_T and the strict upper part is L_T^-T (its diagonal is
// 1/L_cc and is not stored).  A non-positive pivot zeroes its column.
__device__ __forceinline__ void tile16_potrf_inv(double g[4], int lane) {
  const int r = lane & 15, q = lane >> 4;
  // Elimination with UNSCALED columns (LDL^T style): step c only needs 1/d_c
  // (v_rcp_f64 + 2 Newton steps) on its dependent chain
  //   G[r][c2] -= G[r][c] G[c2][c] / d_c.
  // Column c is final after step c, so the 1/sqrt(d_c) scaling of all 16
  // columns is done afterwards, four independent chains per lane, instead of
  // one serial rsq chain inside every step.
  bool okc[4] = {true, true, true, true};
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int cq = c & 3, cj = c >> 2;
    const double d = readlane_f64(g[cj], c + 16 * cq);
    const bool ok = d > 1e-300;
    const double ds = ok ? d : 1.0;
    double ri = __builtin_amdgcn_rcp(ds);
    ri = fma(fma(-ds, ri, 1.0), ri, ri);
    ri = fma(fma(-ds, ri, 1.0), ri, ri);
    const double rinv = ok ? ri : 0.0;
    const double colv = __shfl(g[cj], r + 16 * cq, 64);
    const double mr = ((r == c) ? 1.0 : colv) * rinv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c2 = 4 * j + q;
      const double lc = __shfl(g[cj], c2 + 16 * cq, 64);
      const bool upd = (c2 > c) && ((r <= c) || (c2 <= r));
      if (upd) g[j] = fma(-mr, lc, g[j]);
    }
    if (q == cq) okc[cj] = ok;
  }
  // scale column c by 1/sqrt(d_c) (v_rsq_f64 + Newton), diagonal = sqrt(d_c)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * j + q;
    const double d = __shfl(g[j], c + 16 * q, 64);  // pivot of this lane's column
    const bool ok = okc[j];
    const double ds = ok ? d : 1.0;
    double y = __builtin_amdgcn_rsq(ds);
    const double hd = 0.5 * ds;
#pragma unroll
    for (int nr = 0; nr < 3; ++nr) y = y * fma(-hd * y, y, 1.5);
    double sq = ds * y;
    sq = fma(fma(-sq, sq, ds), 0.5 * y, sq);
    y = fma(fma(-sq, y, 1.0), y, y);
    g[j] = ok ? ((r == c) ? sq : g[j] * y) : 0.0;
  }
}

// Cholesky of the NB x NB tile held in LDS (Lb[c*LS + r], lower triangle) by
// left-looking 16-column panels; leaves L in Lb and the tile inverses
// E_pp = L_pp^-T in Eb.  All 256 threads of the workgroup call it.
constexpr int LS = NB + 1;
constexpr int ES = 17;
__device__ __forceinline__ void factor_tile_lds(double *Lb, double (*Eb)[16 * ES], int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    // (1) left-looking update of panel p: tile (ti,p) -= sum_kt L(ti,kt) L(p,kt)^T
    if (p > 0) {
      const int ti = p + wv;
      if (ti < NP) {
        v4f64 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = Lb[(16 * p + lk + 4 * g) * LS + 16 * ti + lr];
        for (int kc = 0; kc < 16 * p; kc += 4) {
          const double a = -Lb[(kc + lk) * LS + 16 * p + lr];
          const double b = Lb[(kc + lk) * LS + 16 * ti + lr];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
          Lb[(16 * p + lk + 4 * g) * LS + 16 * ti + lr] = acc[g];
      }
      __syncthreads();
    }
    // (2) factor the diagonal tile (wave 0)
    if (wv == 0) {
      const int r = lr, q = lk;
      double g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * j + q;
        g[j] = (r >= c) ? Lb[(16 * p + c) * LS + 16 * p + r] : 0.0;
      }
      tile16_potrf_inv(g, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * j + q;
        if (r >= c) Lb[(16 * p + c) * LS + 16 * p + r] = g[j];
        if (r < c) Eb[p][r * ES + c] = g[j];
        if (r == c) Eb[p][r * ES + c] = (g[j] > 0.0) ? 1.0 / g[j] : 0.0;
      }
    }
    __syncthreads();
    // (3) TRSM of the tiles below: X = T * E_pp   (waves 1..3)
    if (p < NP - 1) {
      const int ti = p + wv;
      if (wv >= 1 && ti < NP) {
        v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double a = Eb[p][(lk + 4 * g) * ES + lr];
          const double b = Lb[(16 * p + lk + 4 * g) * LS + 16 * ti + lr];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        // all reads of this tile precede the writes within the wave
#pragma unroll
        for (int g = 0; g < 4; ++g)
          Lb[(16 * p + lk + 4 * g) * LS + 16 * ti + lr] = acc[g];
      }
      __syncthreads();
    }
  }
}

#ifdef BA_DENSE_DBG
__device__ long long g_dense_dbg[64];
#define DD_STAMP() { if (threadIdx.x == 0 && blockIdx.x == 0 && t0 == 0 && dd_n < 64) g_dense_dbg[dd_n++] = clock64(); }
#else
#define DD_STAMP()
#endif
__global__ __launch_bounds__(256) void k_chol_diag(const double *L, int ld,
                                                   int t0, double *ws_all,
                                                   const int *done) {
#ifdef BA_DENSE_DBG
  int dd_n = 0;
#endif
  DD_STAMP()
  // one workgroup per diagonal tile of the level (independent tiles)
  const int k0 = (t0 + blockIdx.x) * NB;
  double *ws = ws_all + (size_t)(t0 + blockIdx.x) * kDenseWsPerBlock;
  __shared__ double Lb[NB * LS];      // Lb[c*LS + r]
  __shared__ double Eb[NP][16 * ES];  // Eb[p][k*ES + c] = E_pp[k][c]
  const int tid = threadIdx.x;
  // the block is requested before the `done` word is examined: one memory
  // latency for both instead of two in a row
  double lv[NB * NB / 256];
#pragma unroll
  for (int k = 0; k < NB * NB / 256; ++k) {
    const int e = tid + 256 * k;
    const int c = e / NB, r = e % NB;
    lv[k] = (r >= c) ? L[(size_t)(k0 + c) * ld + k0 + r] : 0.0;
  }
  if (done && *done) return;
#pragma unroll
  for (int k = 0; k < NB * NB / 256; ++k) {
    const int e = tid + 256 * k;
    Lb[(e / NB) * LS + e % NB] = lv[k];
  }
  for (int e = tid; e < NP * 16 * ES; e += 256) (&Eb[0][0])[e] = 0.0;
  __syncthreads();
  DD_STAMP()
  factor_tile_lds(Lb, Eb, tid);
  DD_STAMP()
  for (int e = tid; e < NB * NB; e += 256) {
    const int c = e / NB, r = e % NB;
    ws[e] = (r >= c) ? Lb[c * LS + r] : 0.0;
  }
  for (int e = tid; e < NP * 256; e += 256) {
    const int p = e >> 8, k = (e >> 4) & 15, c = e & 15;
    ws[NB * NB + e] = Eb[p][k * ES + c];
  }
  DD_STAMP()
}
#ifdef BA_DENSE_DBG
extern "C" int ba_debug_read_dense(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dense_dbg), sizeof(long long) * 64);
}
#endif

// ---- step 2: TRSM of the rows below the diagonal block --------------------
// X = A21 * L11^-T by block forward substitution over the four 16-column
// panels, entirely on the matrix cores: one wave per 16 rows,
//   X_p = (A_p - sum_{k<p} X_k L_pk^T) * E_pp.
// Orientation D[m = column][n = row]: an accumulator register is 16
// consecutive rows of one column (128 contiguous bytes), and — because the
// f64 C/D map is row = (lane>>4) + 4*reg — the accumulator of one product is
// already the B operand of the next (k-step g <-> k = (lane>>4) + 4g).
__global__ __launch_bounds__(NP * 64) void k_chol_trsm(double *L, int ld,
                                                   int row_limit, int it0,
                                                   const int *__restrict__ item_t,
                                                   const int *__restrict__ item_I,
                                                   const double *__restrict__ ws_all,
                                                   const int *done) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;  // NP waves
  const int lr = lane & 15, lk = lane >> 4;
  // one workgroup per structurally non-zero NB-row tile (t, I) of the level
  const int t = item_t[it0 + blockIdx.x];
  const int k0 = t * NB;
  const int r0 = item_I[it0 + blockIdx.x] * NB + 16 * wv;
  if (r0 >= row_limit) return;
  const double *ws = ws_all + (size_t)t * kDenseWsPerBlock;
  const double *Ld = ws;
  const double *Et = ws + NB * NB;
  // every operand is requested up front (addresses depend only on the item),
  // then the `done` word is examined: one memory latency for the whole kernel
  v4f64 A0[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      A0[p][g] = L[(size_t)(k0 + 16 * p + lk + 4 * g) * ld + r0 + lr];
  double ld_op[NP * (NP - 1) / 2][4], et_op[NP][4];
  {
    int q = 0;
#pragma unroll
    for (int p = 1; p < NP; ++p)
#pragma unroll
      for (int kq = 0; kq < p; ++kq) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          ld_op[q][g] = -Ld[(16 * kq + lk + 4 * g) * NB + 16 * p + lr];
        ++q;
      }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int g = 0; g < 4; ++g) et_op[p][g] = Et[p * 256 + (lk + 4 * g) * 16 + lr];
  if (done && *done) return;
  v4f64 X[NP];
  int q = 0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    v4f64 acc = A0[p];
#pragma unroll
    for (int kq = 0; kq < p; ++kq) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ld_op[q][g], X[kq][g], acc, 0, 0, 0);
      ++q;
    }
    v4f64 out = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g)
      out = __builtin_amdgcn_mfma_f64_16x16x4f64(et_op[p][g], acc[g], out, 0, 0, 0);
    X[p] = out;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      L[(size_t)(k0 + 16 * p + lk + 4 * g) * ld + r0 + lr] = out[g];
  }
}

// ---- step 3: trailing update on the fp64 matrix cores ---------------------
// TARGET-centric: one workgroup per 64x64 tile (I,J) touched in this level;
// it sums the contributions P_I P_J^T of every source panel of the level that
// reaches it (ascending position: deterministic) and applies them with ONE
// read-modify-write.  4 waves of 32x32 (2x2 MFMA 16x16x4 tiles).
// MFMA orientation: the MFMA "row" index runs over C's COLUMN j and the MFMA
// "column" index (lane&15) over C's ROW i, so that each accumulator register
// is 16 consecutive rows of one column = 128 contiguous bytes in memory.
constexpr int MT = NB / 32;  // 16x16 MFMA tiles per wave and dimension
__global__ __launch_bounds__(256) void k_chol_update(double *L, int ld, int tg0,
                                                     const int *__restrict__ tgt_desc,
                                                     const int *__restrict__ src_t,
                                                     const int *done) {
  const int tg = tg0 + blockIdx.x;
  // inline record: I, J, nsrc, src_begin | first four sources
  const int4 d0 = ((const int4 *)tgt_desc)[2 * tg];
  const int4 d1 = ((const int4 *)tgt_desc)[2 * tg + 1];
  const int dn = done ? *done : 0;
  if (dn) return;
  const int I = d0.x, J = d0.y, nsrc = d0.z;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wi = wv & 1, wj = wv >> 1;
  // four waves, each a (NB/2) x (NB/2) quadrant of the target tile
  const int i0 = I * NB + (NB / 2) * wi, j0 = J * NB + (NB / 2) * wj;
  const int lr = lane & 15, lk = lane >> 4;
  // the target tile is requested together with the first source panel
  v4f64 tv[MT][MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < MT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        tv[m][n][g] = L[(size_t)(j0 + 16 * m + lk + 4 * g) * ld + i0 + 16 * n + lr];
  v4f64 acc[MT][MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < MT; ++n) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < nsrc; ++k) {
    const int st = k == 0 ? d1.x : k == 1 ? d1.y : k == 2 ? d1.z : k == 3 ? d1.w : src_t[d0.w + k];
    const double *P = L + (size_t)st * NB * ld;
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; ++kk) {
      const double *col = P + (size_t)(kk * 4 + lk) * ld;
      double av[MT], bv[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        av[m] = col[j0 + 16 * m + lr];
        bv[m] = col[i0 + 16 * m + lr];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < MT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc[m][n], 0, 0, 0);
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < MT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = j0 + 16 * m + lk + 4 * g;
        const int i = i0 + 16 * n + lr;
        L[(size_t)j * ld + i] = tv[m][n][g] - acc[m][n][g];
      }
}

// ---- fused level: diagonal factorisation + TRSM + outer products ----------
// One workgroup per SOURCE tile p of the level, one launch per level (instead
// of diag / TRSM / update launches, each a dependent round trip):
//   1. A_pp = base tile - its pending contribution tiles;  L_pp, E = chol(A_pp)
//   2. every row tile I of p:  P_I = (A_Ip - pending) L_pp^-T  -> L (for the
//      backward sweep) and LDS
//   3. every pair (a >= c) of row tiles:  contribution tile  P_a P_c^T  -> cbuf
// Nothing is updated in place, so no workgroup ever waits for another one of
// the same launch; the sums are formed by the (single) consumer of each tile
// in ascending contribution id: deterministic.
constexpr int PS = NB + 1;
constexpr int MT2 = NB / 16;
constexpr int RG = 4 / NP;        // row tiles handled at once by the four waves
constexpr int TE = NB * NB / 256; // tile elements per thread
constexpr int PCH = 8;            // pending contributions fetched per batch
__global__ __launch_bounds__(256) void k_chol_level(double *L, int ld, int npad, int t0,
                                                    const int *__restrict__ f_desc,
                                                    const int *__restrict__ rows,
                                                    const int *__restrict__ f_pend,
                                                    double *cbuf, double *ws_all,
                                                    const int *done) {
  __shared__ double Lb[NB * LS];
  __shared__ double Eb[NP][16 * ES];
  __shared__ double Pb[kMaxFusedRows][NB * PS];  // Pb[a][k*PS + row]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
#ifdef BA_DENSE_DBG
  int dd_n = 0;
#endif
  DD_STAMP()
  const int p = t0 + blockIdx.x;
  const int k0 = p * NB;
  const int4 *dq = (const int4 *)(f_desc + 16 * (size_t)p);
  const int4 d0 = dq[0];  // nrow, row_begin, pend_begin, pend_n
  const int4 d1 = dq[1];  // out_base, npairs
  const int4 d2 = dq[2], d3 = dq[3];  // first eight row tiles
  const int dn = done ? *done : 0;
  const int nrow = d0.x;
  if (dn) return;
  // ---- 1. every tile of this column, requested at once: base values ----
  // (dependent-load chain of the whole kernel: record -> bases + pending list
  //  -> contribution tiles)
  double lv[TE];
#pragma unroll
  for (int k = 0; k < TE; ++k) {
    const int e = tid + 256 * k;
    lv[k] = L[(size_t)(k0 + e / NB) * ld + k0 + e % NB];
  }
  for (int a0 = 0; a0 < nrow; a0 += 4) {  // four row tiles in flight
    double rv[4][TE];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int a = a0 + j;
      const int I = a >= nrow ? -1
                  : a == 0 ? d2.x : a == 1 ? d2.y : a == 2 ? d2.z : a == 3 ? d2.w
                  : a == 4 ? d3.x : a == 5 ? d3.y : a == 6 ? d3.z : a == 7 ? d3.w
                  : rows[d0.y + a];
#pragma unroll
      for (int k = 0; k < TE; ++k) {
        const int e = tid + 256 * k;
        const int r = I * NB + e % NB;
        rv[j][k] = (I >= 0 && r < npad + 16) ? L[(size_t)(k0 + e / NB) * ld + r] : 0.0;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int a = a0 + j;
      if (a < nrow) {
#pragma unroll
        for (int k = 0; k < TE; ++k) {
          const int e = tid + 256 * k;
          Pb[a][(e / NB) * PS + e % NB] = rv[j][k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < TE; ++k) {
    const int e = tid + 256 * k;
    Lb[(e / NB) * LS + e % NB] = lv[k];
  }
  for (int e = tid; e < NP * 16 * ES; e += 256) (&Eb[0][0])[e] = 0.0;
  DD_STAMP()
  // ---- pending contributions, PCH tiles in flight, subtracted in list order ----
  // (each thread owns the same tile elements for every tile: no barrier needed)
  for (int q0 = 0; q0 < d0.w; q0 += PCH) {
    double cv[PCH][TE];
    int slot[PCH];
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int q = q0 + j < d0.w ? q0 + j : d0.w - 1;
      const int2 pe = ((const int2 *)f_pend)[d0.z + q];
      slot[j] = q0 + j < d0.w ? pe.x : -2;
      const double *C = cbuf + (size_t)pe.y * NB * NB;
#pragma unroll
      for (int k = 0; k < TE; ++k) cv[j][k] = C[tid + 256 * k];
    }
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      if (slot[j] == -2) continue;
      double *dst = slot[j] < 0 ? Lb : Pb[slot[j]];
      const int st = slot[j] < 0 ? LS : PS;
#pragma unroll
      for (int k = 0; k < TE; ++k) {
        const int e = tid + 256 * k;
        dst[(e / NB) * st + e % NB] -= cv[j][k];
      }
    }
  }
  __syncthreads();
  DD_STAMP()
  // ---- 2. factor the diagonal tile ----
  factor_tile_lds(Lb, Eb, tid);
  DD_STAMP()
  {
    double *ws = ws_all + (size_t)p * kDenseWsPerBlock;
    for (int e = tid; e < NB * NB; e += 256) {
      const int c = e / NB, r = e % NB;
      ws[e] = (r >= c) ? Lb[c * LS + r] : 0.0;
    }
    for (int e = tid; e < NP * 256; e += 256) {
      const int pp = e >> 8, k = (e >> 4) & 15, c = e & 15;
      ws[NB * NB + e] = Eb[pp][k * ES + c];
    }
  }
  DD_STAMP()
  // ---- 3. TRSM of the row tiles, in place in LDS (wave group wv / NP: tile,
  //         wave wv % NP: 16 of its rows) ----
  for (int a0 = 0; a0 < nrow; a0 += RG) {
    const int a = a0 + wv / NP, w = wv % NP;
    if (a < nrow) {
      const int I = a == 0 ? d2.x : a == 1 ? d2.y : a == 2 ? d2.z : a == 3 ? d2.w
                  : a == 4 ? d3.x : a == 5 ? d3.y : a == 6 ? d3.z : a == 7 ? d3.w
                  : rows[d0.y + a];
      const int r0 = I * NB + 16 * w;
      v4f64 A0[NP];
#pragma unroll
      for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int g = 0; g < 4; ++g) A0[pp][g] = Pb[a][(16 * pp + lk + 4 * g) * PS + 16 * w + lr];
      v4f64 X[NP];
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        v4f64 acc = A0[pp];
#pragma unroll
        for (int kq = 0; kq < pp; ++kq)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const double lo = -Lb[(16 * kq + lk + 4 * g) * LS + 16 * pp + lr];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lo, X[kq][g], acc, 0, 0, 0);
          }
        v4f64 out = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double eo = Eb[pp][(lk + 4 * g) * ES + lr];
          out = __builtin_amdgcn_mfma_f64_16x16x4f64(eo, acc[g], out, 0, 0, 0);
        }
        X[pp] = out;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (r0 < npad + 16) L[(size_t)(k0 + 16 * pp + lk + 4 * g) * ld + r0 + lr] = out[g];
          Pb[a][(16 * pp + lk + 4 * g) * PS + 16 * w + lr] = out[g];
        }
      }
    }
  }
  __syncthreads();
  DD_STAMP()
  // ---- 4. contribution tiles, one wave per pair (a >= c), rows[c] a real tile ----
  for (int k = wv; k < d1.y; k += 4) {
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= k) ++a;
    const int c = k - a * (a + 1) / 2;
    v4f64 acc[MT2][MT2];
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
      for (int n = 0; n < MT2; ++n) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; ++kk) {
      double av[MT2], bv[MT2];
#pragma unroll
      for (int m = 0; m < MT2; ++m) {
        av[m] = Pb[c][(kk * 4 + lk) * PS + 16 * m + lr];
        bv[m] = Pb[a][(kk * 4 + lk) * PS + 16 * m + lr];
      }
#pragma unroll
      for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int n = 0; n < MT2; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[n], acc[m][n], 0, 0, 0);
    }
    double *C = cbuf + (size_t)(d1.x + k) * NB * NB;
#pragma unroll
    for (int m = 0; m < MT2; ++m)
#pragma unroll
      for (int n = 0; n < MT2; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          C[(16 * m + lk + 4 * g) * NB + 16 * n + lr] = acc[m][n][g];
  }
  DD_STAMP()
}

// ---- backward sweep L^T x = z, one launch per level (reverse order) --------
// Left-looking: the workgroup of tile t gathers  w = z_t - sum_I L(I,t)^T x_I
// over the non-zero row tiles I below t (all solved in earlier launches), then
// wave 0 solves the 64x64 diagonal system by block back substitution with the
// tile inverses:  x_p = E_pp (w_p - sum_{u>p} L_up^T x_u),  p = 3..0.
constexpr int BG = 256 / NB;  // row groups of the gather (each NB/BG rows)
constexpr int BR = NB / BG;
__global__ __launch_bounds__(256) void k_chol_back(const double *L, int ld,
                                                   int npad, int t0,
                                                   const int *__restrict__ back_desc,
                                                   const int *__restrict__ rows,
                                                   const double *__restrict__ ws_all,
                                                   double *xc, double *x,
                                                   const int *__restrict__ col_x,
                                                   const int *done) {
  __shared__ double xs[NB];
  __shared__ double part[BG][NB];
  const int tid = threadIdx.x;
  const int t = t0 + blockIdx.x;
  const int k0 = t * NB;
  // inline record: nrow, row_begin | first six row tiles
  const int4 d0 = ((const int4 *)back_desc)[2 * t];
  const int4 d1 = ((const int4 *)back_desc)[2 * t + 1];
  const int dn = done ? *done : 0;
  const double *Ld = ws_all + (size_t)t * kDenseWsPerBlock;
  const double *Et = Ld + NB * NB;
  // wave 0 needs, much later, operands whose addresses depend on nothing but t:
  // they are requested now so that their latency hides behind the gather
  const int i = tid & 15, q = (tid >> 4) & 3;
  double lop[NP][NP][4];  // [p][u][rr], u > p
  double eop[NP][4], zv[NP];
  int xidx = -1;
  if (tid < 64) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int u = p + 1; u < NP; ++u)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          lop[p][u][rr] = Ld[(16 * p + i) * NB + 16 * u + 4 * q + rr];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) eop[p][cc] = Et[p * 256 + i * 16 + 4 * q + cc];
      zv[p] = L[(size_t)(k0 + 16 * p + i) * ld + npad];
    }
    if (tid < NB) xidx = col_x[k0 + tid];
  }
  if (dn) return;
  {
    const int c = tid % NB, qq = tid / NB;
    const double *colp = L + (size_t)(k0 + c) * ld;
    double s = 0.0;
    const int nrow = d0.x;
    for (int a = 0; a < nrow; ++a) {
      const int I = a == 0 ? d0.z : a == 1 ? d0.w : a == 2 ? d1.x : a == 3 ? d1.y
                  : a == 4 ? d1.z : a == 5 ? d1.w : rows[d0.y + a];
      const double *src = colp + I * NB + BR * qq;
      const double *xi = xc + I * NB + BR * qq;
#pragma unroll
      for (int r = 0; r < BR; ++r) s += src[r] * xi[r];
    }
    part[qq][c] = s;
  }
  __syncthreads();
  if (tid < 64) {
#pragma unroll
    for (int p = NP - 1; p >= 0; --p) {
      double acc = 0.0;
#pragma unroll
      for (int u = p + 1; u < NP; ++u)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = 16 * u + 4 * q + rr;
          acc += lop[p][u][rr] * xs[row];
        }
      acc += __shfl_xor(acc, 16, 64);
      acc += __shfl_xor(acc, 32, 64);
      const int c = 16 * p + i;
      double below = part[0][c];
#pragma unroll
      for (int g = 1; g < BG; ++g) below += part[g][c];
      const double wv = (zv[p] - below) - acc;
      double px = 0.0;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c2 = 4 * q + cc;
        px += eop[p][cc] * __shfl(wv, c2, 64);
      }
      px += __shfl_xor(px, 16, 64);
      px += __shfl_xor(px, 32, 64);
      if (q == 0) xs[16 * p + i] = px;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (tid < NB) {
      xc[k0 + tid] = xs[tid];
      if (xidx >= 0) x[xidx] = xs[tid];
    }
  }
}

}  // namespace

void launch_dense_init(double *L, int ld, const int *col_x, const int *zt_I,
                       const int *zt_J, int n_zt, const int *done_flag,
                       hipStream_t s) {
  if (n_zt > 0)
    BA_LAUNCH(K_DENSE_INIT, k_dense_init, dim3(n_zt), dim3(256), s, L, ld, col_x,
                       zt_I, zt_J, done_flag);
}

// Level-scheduled, structure-aware blocked Cholesky (see ba_dense_sched.h):
// per level one batched diagonal launch, one batched TRSM launch and one
// target-centric update launch; then one backward launch per level.
void dense_factor_solve(double *L, int npad, int ld, double *Ldiag, double *x,
                        const int *done, const DenseSchedule &sc,
                        const DenseDev &dd, hipStream_t s) {
  const int row_limit = npad + 16;  // rows that carry data (rhs = row npad)
  // The fused one-launch-per-level path is opt-in (BA_DENSE_FUSED=1): on C4 it
  // measured no faster than the three-kernel path (886 vs 876 us per LM
  // iteration) because tiles that survive many levels collect long pending
  // lists, which their single consumer then gathers serially.
  const char *fz = getenv("BA_DENSE_FUSED");
  const bool fused = sc.fused_ok && dd.f_desc && fz && fz[0] == '1';
  for (int l = 0; l < sc.nlev; ++l) {
    const int t0 = sc.lev_ptr[l], nt = sc.lev_ptr[l + 1] - t0;
    if (fused) {
      BA_LAUNCH(K_CHOL_LEVEL, k_chol_level, dim3(nt), dim3(256), s, L, ld, npad, t0,
                dd.f_desc, dd.rows, dd.f_pend, dd.cbuf, Ldiag, done);
      continue;
    }
    BA_LAUNCH(K_CHOL_DIAG, k_chol_diag, dim3(nt), dim3(256), s, L, ld, t0, Ldiag,
                       done);
    const int it0 = sc.item_ptr[l], ni = sc.item_ptr[l + 1] - it0;
    if (ni > 0)
      BA_LAUNCH(K_CHOL_TRSM, k_chol_trsm, dim3(ni), dim3(NP * 64), s, L, ld,
                         row_limit, it0, dd.item_t, dd.item_I, Ldiag, done);
    const int tg0 = sc.tgt_ptr[l], ng = sc.tgt_ptr[l + 1] - tg0;
    if (ng > 0)
      BA_LAUNCH(K_CHOL_UPDATE, k_chol_update, dim3(ng), dim3(256), s, L, ld, tg0,
                         dd.tgt_desc, dd.src_t, done);
  }
  for (int l = sc.nlev - 1; l >= 0; --l) {
    const int t0 = sc.lev_ptr[l], nt = sc.lev_ptr[l + 1] - t0;
    BA_LAUNCH(K_CHOL_BACK, k_chol_back, dim3(nt), dim3(256), s, L, ld, npad, t0,
                       dd.back_desc, dd.rows, Ldiag, dd.xc, x, dd.col_x, done);
  }
}

void launch_dense_solve(const DevProblem &d, const DenseSchedule &sc,
                        const DenseDev &dd, hipStream_t s) {
  dense_factor_solve(d.L, d.npad, d.ld, d.Ldiag, d.x, &d.ctrl->done, sc, dd, s);
}

}  // namespace ba
