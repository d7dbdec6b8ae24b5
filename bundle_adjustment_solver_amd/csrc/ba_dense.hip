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
#include "ba_tile16.h"

#include <cstdlib>
#include <vector>

namespace ba {

namespace {
typedef double v4f64 __attribute__((ext_vector_type(4)));
// Tile order: the kernels exist for 32- and 64-column tiles (ba_dense_tile.inc is
// compiled once per order); ba_finalize picks the order per problem from the
// two level schedules.

// (Re)initialise the tiles of L that the factorisation touches: the
// structurally non-zero tiles of the factor (incl. fill-in), the diagonal tiles
// (unit diagonal on padding columns) and the rhs row block.  Every other tile
// was zeroed once at ba_finalize and is never written.
__global__ __launch_bounds__(256) void k_dense_init(double *L, int ld,
                                                    const int *__restrict__ col_x,
                                                    const int *__restrict__ zt_I,
                                                    const int *__restrict__ zt_J,
                                                    int nb, const int *done) {
  if (done && *done) return;
  const int I = zt_I[blockIdx.x], J = zt_J[blockIdx.x];
  for (int e = threadIdx.x; e < nb * nb; e += 256) {
    const int c = J * nb + e / nb, r = I * nb + e % nb;
    L[(size_t)c * ld + r] = (r == c && col_x[c] < 0) ? 1.0 : 0.0;
  }
}

using tile16::readlane_f64;
// -DBA_TILE16_OLD: the first form of the tile factorisation (developer comparison)
#ifdef BA_TILE16_OLD
#define BA_TILE16_POTRF tile16::tile16_potrf_inv
#elif defined(BA_TILE16_CALL)
// ONE copy of the routine per kernel instead of one per (unrolled) panel: the second
// panel of a tile finds its ~8 KB of straight-line code in the instruction cache
__device__ __attribute__((noinline)) int tile16_potrf_call(double g[4], int lane, double &dinv) {
  return tile16::tile16_potrf_inv2(g, lane, dinv);
}
#define BA_TILE16_POTRF tile16_potrf_call
#else
#define BA_TILE16_POTRF tile16::tile16_potrf_inv2
#endif
// dropped pivots are counted per handle (ba_get_dropped_pivots); the integer
// atomic runs only when a factorisation actually meets one
__device__ __forceinline__ void count_bad_pivots(int *bad, int n, int lane) {
  if (bad && n > 0 && lane == 0) atomicAdd(bad, n);
}

#ifdef BA_DENSE_DBG
__device__ long long g_dense_dbg[64];
#define DD_STAMP() { if (threadIdx.x == 0 && blockIdx.x == 0 && t0 == 0 && dd_n < 64) g_dense_dbg[dd_n++] = clock64(); }
#else
#define DD_STAMP()
#endif
namespace nb32 {
constexpr int NB = 32;
#include "ba_dense_tile.inc"
}  // namespace nb32
namespace nb64 {
constexpr int NB = 64;
#include "ba_dense_tile.inc"
}  // namespace nb64
#ifdef BA_DENSE_DBG
extern "C" int ba_debug_read_dense(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dense_dbg), sizeof(long long) * 64);
}
#endif


// ---- the last levels in ONE workgroup ---------------------------------------
// The level schedule ends with levels of 3, 2, 1 tiles, each of which costs a
// full launch + global round-trip chain (~27 us) for almost no work.  The
// trailing block of the dense image (the tiles of the last levels are its last
// columns: positions are in elimination order) is at that point the Schur
// complement of everything eliminated before, with its right-hand side in the
// rhs row.  This kernel takes the whole block (<= kTailCols columns + the 16-row
// rhs block) into LDS, factors it by left-looking 16-column panels exactly like
// factor_tile_lds (MFMA panel update and TRSM, wave 0 factors the 16x16
// diagonal tiles), forward-substitutes the rhs as one more row tile, and runs
// the block back substitution in the same launch; only x leaves.
constexpr int kTailCols = 96;
constexpr int kTailLS = kTailCols + 16 + 1;  // column stride of the LDS image (rows + rhs block + pad)
constexpr int kTailES = 17;
// NPt = 16-column panels of the block (compile time: the panel loops unroll and
// their LDS reads pipeline; with run-time bounds the kernel was twice as slow).
template <int NPt, bool PAIR>
__global__ __launch_bounds__(256) void k_chol_tail(const double *L, int ld, int npad, int c0,
                                                   double *xc, double *x,
                                                   const int *__restrict__ col_x, const int *done,
                                                   int *bad) {
  constexpr int nbt = 16 * NPt;
  static_assert(nbt <= kTailCols, "tail block does not fit the LDS image");
  static_assert(!PAIR || NPt >= 4, "a paired first level has two 32-column tiles");
  __shared__ double Lb[kTailCols * kTailLS];             // Lb[c*LS + r], r < nbt: matrix, r >= nbt: rhs block
  __shared__ double Eb[kTailCols / 16][16 * kTailES];    // Eb[p][k*ES + c] = E_pp[k][c], E_pp = L_pp^-T
  __shared__ double xs[kTailCols];
  constexpr int LS = kTailLS, ES = kTailES;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  constexpr int nr = nbt + 16;  // rows of the LDS image; row tile NPt is the rhs block
  // the whole block is requested before anything waits (one load per
  // iteration followed by its LDS store would pay ~40 memory latencies in a row)
  constexpr int kLd = (nbt * nr + 255) / 256;
  double lv[kLd];
#pragma unroll
  for (int k = 0; k < kLd; ++k) {
    const int e = tid + 256 * k;
    const int c = e / nr, rr = e - c * nr;
    const int row = rr < nbt ? c0 + rr : npad + (rr - nbt);
    lv[k] = (e < nbt * nr && rr >= c) ? L[(size_t)(c0 + c) * ld + row] : 0.0;
  }
  if (done && *done) return;
#pragma unroll
  for (int k = 0; k < kLd; ++k) {
    const int e = tid + 256 * k;
    const int c = e / nr, rr = e - c * nr;
    if (e < nbt * nr) Lb[c * LS + rr] = lv[k];
  }
  for (int e = tid; e < (kTailCols / 16) * 16 * ES; e += 256) (&Eb[0][0])[e] = 0.0;
  __syncthreads();
  // Macro-steps.  PAIR: the first level of the block has TWO 32-column tiles
  // (panels 0,1 and 2,3).  Tiles of one level are independent (the tile between
  // them is structurally zero), so their panels are processed side by side:
  // {0,2}, {1,3}, then 4, 5 — four sequential 16x16 factorisations instead of six.
  constexpr int NG = PAIR ? NPt - 2 : NPt;
#pragma unroll
  for (int m = 0; m < NG; ++m) {
    const int p0 = PAIR ? (m < 2 ? m : m + 2) : m;
    const bool two = PAIR && m < 2;
    // this wave's panel: waves 0,1 -> p0 and waves 2,3 -> p0 + 2 in a paired step
    const int p = (two && wv >= 2) ? p0 + 2 : p0;
    const int w2 = two ? (wv & 1) : wv, nw = two ? 2 : 4;
    // columns that can contribute to panel p: from its own tile on in a paired
    // step (the other tile's columns are zero in these rows), else all earlier ones
    const int kc0 = two ? 32 * (p >> 1) : 0;
    // (1) left-looking update: tile (ti,p) -= sum_kt L(ti,kt) L(p,kt)^T, ti = p .. NPt
    if (m > 0) {
      for (int ti = p + w2; ti <= NPt; ti += nw) {
        v4f64 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = Lb[(16 * p + lk + 4 * g) * LS + 16 * ti + lr];
        for (int kc = kc0; kc < 16 * p; kc += 4) {
          const double a = -Lb[(kc + lk) * LS + 16 * p + lr];
          const double b = Lb[(kc + lk) * LS + 16 * ti + lr];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) Lb[(16 * p + lk + 4 * g) * LS + 16 * ti + lr] = acc[g];
      }
      __syncthreads();
    }
    // (2) factor the diagonal tile(s): wave 0 (and wave 2 for the second panel of a paired step)
    if (wv == 0 || (two && wv == 2)) {
      const int r = lr, q = lk;
      double g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * j + q;
        g[j] = (r >= c) ? Lb[(16 * p + c) * LS + 16 * p + r] : 0.0;
      }
      double dinv;
      count_bad_pivots(bad, BA_TILE16_POTRF(g, lane, dinv), lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * j + q;
        if (r >= c) Lb[(16 * p + c) * LS + 16 * p + r] = g[j];
        if (r < c) Eb[p][r * ES + c] = g[j];
        if (r == c) Eb[p][r * ES + c] = dinv;
      }
    }
    __syncthreads();
    // (3) TRSM of the tiles below (incl. the rhs block): X = T * E_pp
    for (int ti = p + 1 + w2; ti <= NPt; ti += nw) {
      v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const double a = Eb[p][(lk + 4 * g) * ES + lr];
        const double b = Lb[(16 * p + lk + 4 * g) * LS + 16 * ti + lr];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
      // all reads of this tile precede the writes within the wave
#pragma unroll
      for (int g = 0; g < 4; ++g) Lb[(16 * p + lk + 4 * g) * LS + 16 * ti + lr] = acc[g];
    }
    __syncthreads();
  }
  // (4) L^T x = y by block back substitution (wave 0): y is row 0 of the rhs block,
  //     x_p = E_pp (y_p - sum_{u>p} L_up^T x_u)
  if (wv == 0) {
    const int i = lr, q = lk;
#pragma unroll
    for (int p = NPt - 1; p >= 0; --p) {
      double acc = 0.0;
#pragma unroll
      for (int u = p + 1; u < NPt; ++u)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = 16 * u + 4 * q + rr;
          acc += Lb[(16 * p + i) * LS + row] * xs[row];
        }
      acc += __shfl_xor(acc, 16, 64);
      acc += __shfl_xor(acc, 32, 64);
      const double wvv = Lb[(16 * p + i) * LS + nbt] - acc;
      double px = 0.0;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c2 = 4 * q + cc;
        px += Eb[p][i * ES + c2] * __shfl(wvv, c2, 64);
      }
      px += __shfl_xor(px, 16, 64);
      px += __shfl_xor(px, 32, 64);
      if (q == 0) xs[16 * p + i] = px;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  if (tid < nbt) {
    xc[c0 + tid] = xs[tid];
    const int xi = col_x[c0 + tid];
    if (xi >= 0) x[xi] = xs[tid];
  }
}

}  // namespace

void launch_dense_init(double *L, int ld, const int *col_x, const int *zt_I,
                       const int *zt_J, int n_zt, int nb, const int *done_flag,
                       hipStream_t s) {
  if (n_zt > 0)
    BA_LAUNCH(K_DENSE_INIT, k_dense_init, dim3(n_zt), dim3(256), s, L, ld, col_x,
                       zt_I, zt_J, nb, done_flag);
}

// Level-scheduled, structure-aware blocked Cholesky (see ba_dense_sched.h):
// per level one batched diagonal launch, one batched TRSM launch and one
// target-centric update launch; then one backward launch per level.
// NS = nb32 or nb64 (the kernels of the schedule's tile order).
#define BA_DENSE_RUN(NS)                                                                    \
  if (look) {                                                                               \
    /* LOOKAHEAD (dense patterns, three-kernel path): the targets of level l that lie in a  \
       column of level l + 1 are updated first; the factorisation + TRSM of level l + 1 then \
       run on the auxiliary stream BESIDE the bulk of level l's update */                   \
    const int nlv = sc.nlev - tail_levels;                                                  \
    hipStream_t X = dd.aux_stream;                                                          \
    for (int l = 0; l < nlv; ++l) {                                                         \
      const int tg0 = sc.tgt_ptr[l], ng = sc.tgt_ptr[l + 1] - tg0, nf = sc.tgt_first[l];    \
      if (l == 0) {                                                                         \
        const int t0 = sc.lev_ptr[0], nt = sc.lev_ptr[1] - t0;                              \
        const int it0 = sc.item_ptr[0], ni = sc.item_ptr[1] - it0;                          \
        BA_LAUNCH(K_CHOL_DIAG, NS::k_chol_diag, dim3(nt), dim3(256), s, L, ld, t0, Ldiag, done, bad); \
        if (ni > 0)                                                                         \
          BA_LAUNCH(K_CHOL_TRSM, NS::k_chol_trsm, dim3(ni), dim3(NS::NP * 64), s, L, ld,    \
                    row_limit, it0, dd.item_t, dd.item_I, Ldiag, done);                     \
      } else {                                                                              \
        (void)hipStreamWaitEvent(s, dd.ev_x[l & 1], 0);                                     \
      }                                                                                     \
      if (nf > 0)                                                                           \
        BA_LAUNCH(K_CHOL_UPDATE, NS::k_chol_update, dim3(nf), dim3(256), s, L, ld, tg0,     \
                  dd.tgt_desc, dd.src_t, done);                                             \
      if (l + 1 < nlv) {                                                                    \
        const int t1 = sc.lev_ptr[l + 1], nt1 = sc.lev_ptr[l + 2] - t1;                     \
        const int it1 = sc.item_ptr[l + 1], ni1 = sc.item_ptr[l + 2] - it1;                 \
        (void)hipEventRecord(dd.ev_m, s);                                                   \
        (void)hipStreamWaitEvent(X, dd.ev_m, 0);                                            \
        BA_LAUNCH(K_CHOL_DIAG, NS::k_chol_diag, dim3(nt1), dim3(256), X, L, ld, t1, Ldiag, done, bad); \
        if (ni1 > 0)                                                                        \
          BA_LAUNCH(K_CHOL_TRSM, NS::k_chol_trsm, dim3(ni1), dim3(NS::NP * 64), X, L, ld,   \
                    row_limit, it1, dd.item_t, dd.item_I, Ldiag, done);                     \
        (void)hipEventRecord(dd.ev_x[(l + 1) & 1], X);                                      \
      }                                                                                     \
      if (ng - nf > 0)                                                                      \
        BA_LAUNCH(K_CHOL_UPDATE, NS::k_chol_update, dim3(ng - nf), dim3(256), s, L, ld,     \
                  tg0 + nf, dd.tgt_desc, dd.src_t, done);                                   \
    }                                                                                       \
  } else if (look2) {                                                                       \
    /* dense patterns: lookahead inside one launch per level (k_chol_look) */               \
    const int nlv = sc.nlev - tail_levels;                                                  \
    {                                                                                       \
      const int t0 = sc.lev_ptr[0], nt = sc.lev_ptr[1] - t0;                                \
      const int it0 = sc.item_ptr[0], ni = sc.item_ptr[1] - it0;                            \
      BA_LAUNCH(K_CHOL_DIAG, NS::k_chol_diag, dim3(nt), dim3(256), s, L, ld, t0, Ldiag, done, bad); \
      if (ni > 0)                                                                           \
        BA_LAUNCH(K_CHOL_TRSM, NS::k_chol_trsm, dim3(ni), dim3(NS::NP * 64), s, L, ld,      \
                  row_limit, it0, dd.item_t, dd.item_I, Ldiag, done);                       \
    }                                                                                       \
    for (int l = 0; l < nlv; ++l) {                                                         \
      const int tg0 = sc.tgt_ptr[l], ng = sc.tgt_ptr[l + 1] - tg0, nf = sc.tgt_first[l];    \
      if (l + 1 < nlv) {                                                                    \
        const int t1 = sc.lev_ptr[l + 1], nt1 = sc.lev_ptr[l + 2] - t1;                     \
        const int it1 = sc.item_ptr[l + 1], ni1 = sc.item_ptr[l + 2] - it1;                 \
        BA_LAUNCH(K_CHOL_UPDATE, NS::k_chol_look, dim3(nt1 + ni1 + ng), dim3(256), s, L,    \
                  ld, row_limit, nf, t1, nt1, it1, ni1, tg0, dd.item_t, dd.item_I, Ldiag,   \
                  dd.tgt_desc, dd.src_t, dd.look_need, done, bad, dd.dag_dflags,            \
                  dd.fwd_cnt, gen_now);                                                     \
      } else if (ng > 0) {                                                                  \
        BA_LAUNCH(K_CHOL_UPDATE, NS::k_chol_update, dim3(ng), dim3(256), s, L, ld, tg0,     \
                  dd.tgt_desc, dd.src_t, done);                                             \
      }                                                                                     \
    }                                                                                       \
  } else if (dag) {                                                                         \
    /* three-kernel path: every non-tail level in ONE dataflow launch with lookahead */     \
    BA_LAUNCH(K_CHOL_UPDATE, NS::k_chol_dag, dim3(dd.n_dag_items), dim3(256), s, L, ld,     \
              row_limit, (const int2 *)dd.dag_items, dd.n_dag_items, dd.item_t, dd.item_I,  \
              dd.dag_ntrsm, Ldiag, dd.tgt_desc, dd.src_t, dd.upd_pre, dd.col_need, done,    \
              bad, dd.fwd_flags, dd.dag_dflags, dd.dag_tcnt, dd.fwd_cnt, dd.fwd_ticket,     \
              gen_now);                                                                     \
  } else if (fwd_flow) {                                                                    \
    /* every non-tail level — factorisation + TRSM and updates — in ONE dataflow launch */  \
    BA_LAUNCH(K_CHOL_DIAG_TRSM, NS::k_chol_fwd_flow, dim3(dd.n_fwd_items), dim3(256), s, L, \
              ld, npad, (const int2 *)dd.fwd_items, dd.n_fwd_items, dd.row_desc, dd.rows,   \
              Ldiag, dd.tgt_desc, dd.src_t, dd.upd_pre, dd.col_need, done, bad,             \
              dd.fwd_flags, dd.fwd_cnt,                                                     \
              (dd.n_fwd_items <= kFlowResident && !dd.force_ticket) ? nullptr : dd.fwd_ticket, \
              gen_now);                                                                     \
  } else                                                                                    \
  for (int l = 0; l < sc.nlev - tail_levels; ++l) {                                         \
    const int t0 = sc.lev_ptr[l], nt = sc.lev_ptr[l + 1] - t0;                              \
    if (fused) {                                                                            \
      BA_LAUNCH(K_CHOL_LEVEL, NS::k_chol_level, dim3(nt), dim3(256), s, L, ld, npad, t0,    \
                dd.f_desc, dd.rows, dd.f_pend, dd.cbuf, Ldiag, done, bad);                       \
      continue;                                                                             \
    }                                                                                       \
    const int it0 = sc.item_ptr[l], ni = sc.item_ptr[l + 1] - it0;                          \
    if (split) {                                                                            \
      BA_LAUNCH(K_CHOL_DIAG, NS::k_chol_diag, dim3(nt), dim3(256), s, L, ld, t0, Ldiag, done, bad); \
      if (ni > 0)                                                                           \
        BA_LAUNCH(K_CHOL_TRSM, NS::k_chol_trsm, dim3(ni), dim3(NS::NP * 64), s, L, ld,      \
                  row_limit, it0, dd.item_t, dd.item_I, Ldiag, done);                       \
    } else if (flow && dd.fwd_flags) {                                                      \
      /* factorisation + TRSM and the updates that consume them: ONE dataflow launch */     \
      const int tg0f = sc.tgt_ptr[l], ngf = sc.tgt_ptr[l + 1] - tg0f;                       \
      BA_LAUNCH(K_CHOL_DIAG_TRSM, NS::k_chol_level_flow, dim3(nt + ngf), dim3(256), s, L,   \
                ld, npad, t0, nt, tg0f, ngf, dd.row_desc, dd.rows, Ldiag, dd.tgt_desc,      \
                dd.src_t, done, bad, dd.fwd_flags,                                          \
                (nt + ngf <= kFlowResident && !dd.force_ticket) ? nullptr : dd.fwd_ticket, gen_now); \
      continue;                                                                             \
    } else {                                                                                \
      BA_LAUNCH(K_CHOL_DIAG_TRSM, NS::k_chol_diag_trsm, dim3(nt), dim3(256), s, L, ld,      \
                npad, t0, dd.row_desc, dd.rows, Ldiag, done, bad);                              \
    }                                                                                       \
    const int tg0 = sc.tgt_ptr[l], ng = sc.tgt_ptr[l + 1] - tg0;                            \
    if (ng > 0)                                                                             \
      BA_LAUNCH(K_CHOL_UPDATE, NS::k_chol_update, dim3(ng), dim3(256), s, L, ld, tg0,       \
                dd.tgt_desc, dd.src_t, done);                                               \
  }                                                                                         \
  if (tail_cols == 64)                                                                      \
    BA_LAUNCH(K_CHOL_TAIL, (k_chol_tail<4, false>), dim3(1), dim3(256), s, L, ld, npad,     \
              tail_c0, dd.xc, x, dd.col_x, done, bad);                                        \
  if (tail_cols == 96 && !tail_pair)                                                        \
    BA_LAUNCH(K_CHOL_TAIL, (k_chol_tail<6, false>), dim3(1), dim3(256), s, L, ld, npad,     \
              tail_c0, dd.xc, x, dd.col_x, done, bad);                                        \
  if (tail_cols == 96 && tail_pair)                                                         \
    BA_LAUNCH(K_CHOL_TAIL, (k_chol_tail<6, true>), dim3(1), dim3(256), s, L, ld, npad,      \
              tail_c0, dd.xc, x, dd.col_x, done, bad);                                        \
  if (flow && !flow_back && n_back > 0) {                                                   \
    BA_LAUNCH(K_CHOL_BACK, NS::k_chol_back_flow<true>, dim3(n_back), dim3(256), s, L, ld,   \
              npad, dd.flow_order, n_back, back_t_end, dd.back_desc, dd.rows, Ldiag, dd.xc, \
              x, dd.col_x, done, dd.flow_flags,                                             \
              (n_back <= kFlowResident && !dd.force_ticket) ? nullptr : dd.flow_ticket, gen_now, bad, \
              (fwd_flow || dag || look2) ? dd.fwd_cnt : nullptr, dd.n_fwd_cnt);                               \
  } else if (flow_back && n_back > 0) {                                                     \
    BA_LAUNCH(K_CHOL_BACK, NS::k_chol_back_flow<false>, dim3(n_back), dim3(256), s, L, ld, npad, \
              dd.flow_order, n_back, back_t_end, dd.back_desc, dd.rows, Ldiag, dd.xc, x,    \
              dd.col_x, done, dd.flow_flags,                                                \
              (n_back <= kFlowResident && !dd.force_ticket) ? nullptr : dd.flow_ticket, gen_now, bad, \
              (fwd_flow || dag || look2) ? dd.fwd_cnt : nullptr, dd.n_fwd_cnt);                               \
  } else                                                                                    \
  for (int l = sc.nlev - tail_levels - 1; l >= 0; --l) {                                    \
    const int t0 = sc.lev_ptr[l], nt = sc.lev_ptr[l + 1] - t0;                              \
    BA_LAUNCH(K_CHOL_BACK, NS::k_chol_back, dim3(nt), dim3(256), s, L, ld, npad, t0,        \
              dd.back_desc, dd.rows, Ldiag, dd.xc, x, dd.col_x, done);                      \
  }

// Workgroups of a dataflow launch that are certainly resident at once on the part (256
// CUs x 2 workgroups of 256 threads at the kernels' register budgets, with a margin):
// up to here the role is the block index, beyond it a ticket (ba_dense_tile.inc).
constexpr int kFlowResident = 448;

// Levels handed to k_chol_tail (the last ones, together 64 or 96 columns, at least two).
static int dense_tail_levels(const DenseSchedule &sc, const DenseDev &dd, bool fused, int *cols_out) {
  int tail_levels = 0, tail_cols = 0;
  if (dd.want_tail && !fused) {
    for (int l = sc.nlev - 1; l >= 0; --l) {
      const int cols = (sc.lev_ptr[l + 1] - sc.lev_ptr[l]) * sc.nb;
      if (tail_cols + cols > kTailCols) break;
      tail_cols += cols;
      ++tail_levels;
    }
    if (tail_levels < 2 || (tail_cols != 64 && tail_cols != 96)) tail_levels = tail_cols = 0;
  }
  if (cols_out) *cols_out = tail_cols;
  return tail_levels;
}
// Positions of the backward sweep's dataflow launch, top level first; returns the first
// position of the tail block (= their number).
int dense_flow_order(const DenseSchedule &sc, const DenseDev &dd, std::vector<int> &order) {
  const bool fused = sc.fused_ok && dd.f_desc && dd.want_fused;
  const int tail_levels = dense_tail_levels(sc, dd, fused, nullptr);
  order.clear();
  for (int l = sc.nlev - tail_levels - 1; l >= 0; --l)
    for (int t = sc.lev_ptr[l]; t < sc.lev_ptr[l + 1]; ++t) order.push_back(t);
  return sc.lev_ptr[sc.nlev - tail_levels];
}

bool dense_fwd_items(const DenseSchedule &sc, const DenseDev &dd, std::vector<int> &items,
                     std::vector<int> &pre, std::vector<int> &need) {
  const bool fused = sc.fused_ok && dd.f_desc && dd.want_fused;
  const bool split = dd.want_split || !dd.row_desc || sc.max_rows > 4 * (4 / (sc.nb / 16));
  const int tail_levels = dense_tail_levels(sc, dd, fused, nullptr);
  const int nlv = sc.nlev - tail_levels;
  items.clear();
  pre.assign(std::max<size_t>(1, sc.tgt_J.size()), 0);
  need.assign((size_t)sc.ncb + 1, 0);
  if (fused || split || nlv < 2) return false;
  auto push = [&](int kind, int id) {
    items.push_back(kind);
    items.push_back(id);
  };
  // LOOKAHEAD order (as k_chol_dag): the targets of level l in a column of level l + 1 first,
  // then the tiles of level l + 1, then the rest of level l's targets
  const bool ahead = (int)sc.tgt_first.size() >= sc.nlev;
  for (int t = sc.lev_ptr[0]; t < sc.lev_ptr[1]; ++t) push(0, t);
  for (int l = 0; l < nlv; ++l) {
    const int tg0 = sc.tgt_ptr[l], tg1 = sc.tgt_ptr[l + 1], nf = ahead ? sc.tgt_first[l] : tg1 - tg0;
    // (targets of one level on one column do not wait for each other: count after the level)
    for (int tg = tg0; tg < tg1; ++tg) pre[tg] = need[sc.tgt_J[tg]];
    for (int tg = tg0; tg < tg0 + nf; ++tg) push(1, tg);
    if (l + 1 < nlv)
      for (int t = sc.lev_ptr[l + 1]; t < sc.lev_ptr[l + 2]; ++t) push(0, t);
    for (int tg = tg0 + nf; tg < tg1; ++tg) push(1, tg);
    for (int tg = tg0; tg < tg1; ++tg) ++need[sc.tgt_J[tg]];
  }
  return true;
}

bool dense_dag_items(const DenseSchedule &sc, const DenseDev &dd, std::vector<int> &items,
                     std::vector<int> &pre, std::vector<int> &need, std::vector<int> &ntrsm,
                     std::vector<int> &look_need) {
  const bool fused = sc.fused_ok && dd.f_desc && dd.want_fused;
  const bool split = dd.want_split || !dd.row_desc || sc.max_rows > 4 * (4 / (sc.nb / 16));
  const int tail_levels = dense_tail_levels(sc, dd, fused, nullptr);
  const int nlv = sc.nlev - tail_levels;
  items.clear();
  pre.assign(std::max<size_t>(1, sc.tgt_J.size()), 0);
  need.assign((size_t)sc.ncb + 1, 0);
  ntrsm.assign((size_t)sc.ncb + 1, 0);
  look_need.assign((size_t)sc.ncb + 1, 0);
  if (fused || !split || nlv < 2 || (int)sc.tgt_first.size() < sc.nlev) return false;
  for (size_t q = 0; q < sc.item_t.size(); ++q) ++ntrsm[sc.item_t[q]];
  auto push = [&](int kind, int id) {
    items.push_back(kind);
    items.push_back(id);
  };
  auto tiles_of = [&](int l) {
    for (int t = sc.lev_ptr[l]; t < sc.lev_ptr[l + 1]; ++t) push(0, t);
    for (int q = sc.item_ptr[l]; q < sc.item_ptr[l + 1]; ++q) push(1, q);
  };
  tiles_of(0);
  for (int l = 0; l < nlv; ++l) {
    const int tg0 = sc.tgt_ptr[l], tg1 = sc.tgt_ptr[l + 1], nf = sc.tgt_first[l];
    for (int tg = tg0; tg < tg1; ++tg) pre[tg] = need[sc.tgt_J[tg]];  // (updates of EARLIER levels)
    for (int tg = tg0; tg < tg0 + nf; ++tg) {
      push(2, tg);
      ++look_need[sc.tgt_J[tg]];  // (k_chol_look: the "first" targets of the level before the tile's)
    }
    if (l + 1 < nlv) tiles_of(l + 1);  // the next level's tiles beside the bulk of this level's update
    for (int tg = tg0 + nf; tg < tg1; ++tg) push(2, tg);
    for (int tg = tg0; tg < tg1; ++tg) ++need[sc.tgt_J[tg]];
  }
  return true;
}

void dense_factor_solve(double *L, int npad, int ld, double *Ldiag, double *x,
                        const int *done, const DenseSchedule &sc,
                        const DenseDev &dd, hipStream_t s) {
  int *bad = dd.bad_pivots;
  const int row_limit = npad + 16;  // rows that carry data (rhs = row npad)
  // The fused one-launch-per-level path is opt-in (BA_DENSE_FUSED=1): on C4 it
  // measured no faster than the three-kernel path (886 vs 876 us per LM
  // iteration) because tiles that survive many levels collect long pending
  // lists, which their single consumer then gathers serially.
  // (the knobs are read when the schedule is uploaded, not once per LM iteration)
  const bool fused = sc.fused_ok && dd.f_desc && dd.want_fused;
  // BA_DENSE_SPLIT=1: separate diagonal and TRSM launches (the TRSM then spreads
  // over one workgroup per row tile: better for dense patterns with many row
  // tiles per column); default: the tile's workgroup also solves its row tiles
  // (all row tiles of a column must fit the prefetched passes: 4 passes x
  //  (4 waves / (nb/16)) tiles)
  // (measured: letting the tile's workgroup fetch further passes as it goes — up to twice
  //  the prefetched capacity — and running the level as one dataflow launch LOSES against
  //  the three launches: C1 0.324 vs 0.316 ms, W20 1.148 vs 1.034 ms per iteration)
  const bool split = dd.want_split || !dd.row_desc || sc.max_rows > 4 * (4 / (sc.nb / 16));
  (void)row_limit;
  // the last levels (at least two, together at most kTailCols columns) are
  // handed to k_chol_tail: one launch instead of three per level (BA_DENSE_TAIL=0: off)
  int tail_cols = 0;
  const int tail_levels = dense_tail_levels(sc, dd, fused, &tail_cols);
  const int tail_c0 = tail_levels > 0 ? sc.lev_ptr[sc.nlev - tail_levels] * sc.nb : 0;
  // two (independent) tiles in the first level of the block: their panels go side by side
  const bool tail_pair = tail_levels > 0 && sc.nb == 32 &&
                         sc.lev_ptr[sc.nlev - tail_levels + 1] - sc.lev_ptr[sc.nlev - tail_levels] == 2;
  (void)tail_pair;
  // one dataflow launch for the backward sweep (BA_DENSE_FLOW=0: one launch per level): the
  // non-tail positions [0, back_t_end), uploaded top level first in dd.flow_order
  const int back_t_end = sc.lev_ptr[sc.nlev - tail_levels];
  const int n_back = back_t_end;
  const bool flow = dd.want_flow && dd.flow_ok && dd.flow_order && dd.n_flow == n_back &&
                    dd.flow_tail_t0 == back_t_end && !fused;
  const int gen_now = ++dd.flow_gen;  // generation number of this solve (flags are never reset)
  // Two forms of the dataflow BACKWARD sweep.  Narrow patterns: wait for all row tiles,
  // then gather (their factor tiles prefetched).  Many row tiles per column (dense
  // patterns: DENSE1K, up to ncb of them): that form puts a 3 MB gather behind the last
  // hand-off (5.0 ms against 3.5 ms for one launch per level); the ORDERED form consumes
  // the row tiles one by one as their flags come up.
  const bool flow_back = flow && sc.max_rows <= 12;
  // lookahead over the levels of the three-kernel path (see BA_DENSE_RUN): needs the
  // auxiliary stream and its events; not under per-kernel timing (serial order) or capture
  const bool look = split && !fused && dd.want_look && dd.flow_ok && dd.aux_stream && dd.ev_m &&
                    !(g_ktimer && g_ktimer->on) && sc.nlev - tail_levels >= 3 &&
                    (int)sc.tgt_first.size() >= sc.nlev;
  // the forward sweep of all non-tail levels as one dataflow launch (opt-in, BA_DENSE_FWD_FLOW=1:
  // see DenseDev::want_fwd_flow); its counters are zeroed again by the backward launch
  const bool fwd_flow = flow && !split && !look && dd.want_fwd_flow && dd.fwd_flags && dd.fwd_items &&
                        dd.n_fwd_items > 0 && dd.fwd_cnt && n_back > 0;
  // the three-kernel path (dense patterns) as one dataflow launch with lookahead (BA_DENSE_DAG=0:
  // three launches per level)
  const bool dag = flow && split && !look && dd.want_dag && dd.dag_items && dd.n_dag_items > 0 &&
                   (dd.n_dag_items <= DenseDev::kDagMaxItems || dd.force_dag) && !dd.force_look2 &&
                   dd.fwd_flags && dd.fwd_cnt && dd.dag_dflags && dd.dag_tcnt && n_back > 0;
  // ... and beyond that item count: lookahead inside one launch per level (BA_DENSE_LOOK2=0: off)
  // (its waiting roles take their place from the block index: all of them — first targets, tiles
  //  and TRSM items of a level — must be resident at once, whatever the dispatch order)
  bool look2_fits = true;
  for (int l = 0; l + 1 < sc.nlev - tail_levels && look2_fits; ++l)
    look2_fits = (int)sc.tgt_first.size() > l &&
                 sc.tgt_first[l] + (sc.lev_ptr[l + 2] - sc.lev_ptr[l + 1]) + (sc.item_ptr[l + 2] - sc.item_ptr[l + 1]) <= kFlowResident;
  const bool look2 = flow && split && !look && !dag && dd.want_look2 && look2_fits && dd.dag_dflags && dd.look_need &&
                     dd.fwd_cnt &&
                     sc.nlev - tail_levels >= 3 &&
                     (int)sc.tgt_first.size() >= sc.nlev;
  if (sc.nb == 32) {
    BA_DENSE_RUN(nb32)
  } else {
    BA_DENSE_RUN(nb64)
  }
}

void launch_dense_solve(const DevProblem &d, const DenseSchedule &sc,
                        const DenseDev &dd, hipStream_t s) {
  dense_factor_solve(d.L, d.npad, d.ld, d.Ldiag, d.x, &d.ctrl->done, sc, dd, s);
}

}  // namespace ba
