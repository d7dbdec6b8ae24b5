// ba_tile16.h — factorisation of one 16x16 SPD tile by one wave, in registers:
// L_T and L_T^-T in the same sweep.  Shared by ba_dense.hip (the reduced camera
// solve: every level of the schedule has two of these on its dependent chain) and
// tools/tile16_bench.hip (cycle counts and accuracy of the variants).  Internal.
#ifndef BA_TILE16_H_
#define BA_TILE16_H_

#include <hip/hip_runtime.h>

// tools/tile16_bench.hip defines BA_T16_STAMP(k) to record s_memtime at phase k
#ifndef BA_T16_STAMP
#define BA_T16_STAMP(k)
#endif

namespace ba {
namespace tile16 {
typedef double v4f64 __attribute__((ext_vector_type(4)));

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
// lower triangle is L_T and the strict upper part is L_T^-T; `dinv` receives
// the diagonal of L_T^-T (= 1/L_cc) in the lane that holds G[c][c].  A
// non-positive pivot zeroes its column.  Returns the number of such pivots
// (wave-uniform).
//
// Blocked by 4, with UNSCALED columns (LDL^T style) inside the sweep.  The column
// operations that turn A into L turn I into L^-T (the strict upper part of the
// tile is the folded image of I), so for every 4-column block
//  (1) every lane fetches the 4x4 diagonal block (v_readlane: uniform) and
//      eliminates it redundantly: the dependent chain per pivot is one
//      v_rcp_f64 + 2 Newton steps + two FMAs, with no cross-lane traffic on it;
//  (2) every row applies the block's unit-lower factor to its four block entries
//      (three cross-lane reads, issued before the chain starts);
//  (3) the rank-4 update of ALL trailing columns (rows of L and rows of the
//      folded inverse alike) is ONE v_mfma_f64_16x16x4_f64 (operands y r_k and
//      y), whose accumulator layout (row = lane&15, column = 4*reg + lane>>4) is
//      exactly this register layout.
// The 1/sqrt(d_c) scaling of all 16 columns happens once afterwards, four
// independent chains per lane.  16 dependent column steps of ~460 cycles became
// 4 block steps.
__device__ __forceinline__ double rcp_refined(double d, bool ok) {
  const double ds = ok ? d : 1.0;
  double ri = __builtin_amdgcn_rcp(ds);
  ri = fma(fma(-ds, ri, 1.0), ri, ri);
  ri = fma(fma(-ds, ri, 1.0), ri, ri);
  return ok ? ri : 0.0;
}
__device__ __forceinline__ double sel4(int k, double a0, double a1, double a2, double a3) {
  return k == 0 ? a0 : k == 1 ? a1 : k == 2 ? a2 : a3;
}
__device__ __forceinline__ int tile16_potrf_inv(double g[4], int lane, double &dinv) {
  const int r = lane & 15, q = lane >> 4;
  int nbad = 0;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int c0 = 4 * kb;
    BA_T16_STAMP(5 * kb + 0)
    // this row's entries of the block columns (own value for k == q)
    const double a0 = __shfl(g[kb], r, 64), a1 = __shfl(g[kb], r + 16, 64),
                 a2 = __shfl(g[kb], r + 32, 64), a3 = __shfl(g[kb], r + 48, 64);
    // diagonal block, lower part: D[i][k] lives in lane (c0 + i, k)
    const double d00 = readlane_f64(g[kb], c0 + 0);
    const double w10 = readlane_f64(g[kb], c0 + 1), d11 = readlane_f64(g[kb], c0 + 1 + 16);
    const double w20 = readlane_f64(g[kb], c0 + 2), d21 = readlane_f64(g[kb], c0 + 2 + 16),
                 d22 = readlane_f64(g[kb], c0 + 2 + 32);
    const double w30 = readlane_f64(g[kb], c0 + 3), d31 = readlane_f64(g[kb], c0 + 3 + 16),
                 d32 = readlane_f64(g[kb], c0 + 3 + 32), d33 = readlane_f64(g[kb], c0 + 3 + 48);
    BA_T16_STAMP(5 * kb + 1)
    // unit-lower LDL^T of the 4x4 block: w = unscaled column entries, l = w / pivot
    const bool ok0 = d00 > 1e-300;
    const double r0 = rcp_refined(d00, ok0);
    const double l10 = w10 * r0, l20 = w20 * r0, l30 = w30 * r0;
    const double p1 = fma(-l10, w10, d11);
    const double w21 = fma(-l20, w10, d21), w31 = fma(-l30, w10, d31);
    const bool ok1 = p1 > 1e-300;
    const double r1 = rcp_refined(p1, ok1);
    const double l21 = w21 * r1, l31 = w31 * r1;
    const double p2 = fma(-l21, w21, fma(-l20, w20, d22));
    const double w32 = fma(-l31, w21, fma(-l30, w20, d32));
    const bool ok2 = p2 > 1e-300;
    const double r2 = rcp_refined(p2, ok2);
    const double l32 = w32 * r2;
    const double p3 = fma(-l32, w32, fma(-l31, w31, fma(-l30, w30, d33)));
    const bool ok3 = p3 > 1e-300;
    const double r3 = rcp_refined(p3, ok3);
    nbad += (ok0 ? 0 : 1) + (ok1 ? 0 : 1) + (ok2 ? 0 : 1) + (ok3 ? 0 : 1);
    BA_T16_STAMP(5 * kb + 2)
    // y = a Ltilde^-T: the unscaled column entries of this row (rows below the
    // block: L times sqrt(d); rows above it: the folded inverse, same transform;
    // block rows: at and below the diagonal y reproduces their w / pivots)
    const double y0 = a0;
    const double y1 = fma(-y0, l10, a1);
    const double y2 = fma(-y1, l21, fma(-y0, l20, a2));
    const double y3 = fma(-y2, l32, fma(-y1, l31, fma(-y0, l30, a3)));
    // above the diagonal a block row holds its folded identity row
    // e_rb Ltilde^-T (unit upper triangular)
    const int rb = r - c0;  // 0..3 inside the block
    const bool inb = rb >= 0 && rb < 4;
    const double e0 = rb == 0 ? 1.0 : 0.0, e1 = rb == 1 ? 1.0 : 0.0, e2 = rb == 2 ? 1.0 : 0.0;
    const double u1 = fma(-e0, l10, e1);
    const double u2 = fma(-u1, l21, fma(-e0, l20, e2));
    const double u3 = fma(-u2, l32, fma(-u1, l31, fma(-e0, l30, rb == 3 ? 1.0 : 0.0)));
    const double ya = sel4(q, y0, y1, y2, y3);
    const double yu = sel4(q, e0, u1, u2, u3);
    const double rq = sel4(q, r0, r1, r2, r3);
    g[kb] = (inb && q > rb) ? yu : ya;
    BA_T16_STAMP(5 * kb + 3)
    // rank-4 update of the trailing columns c2 > c0 + 3:
    //   G[row][c2] -= sum_k B[row][k] * (y[c2][k] r_k)
    // A operand: the panel below the block, scaled by the pivots' reciprocals;
    // B operand: the panel as every ROW sees it (rows below: y; folded inverse
    // rows above: y; block rows: their folded identity row, zero left of its 1)
    if (kb < 3) {
      const double aop = r > c0 + 3 ? -ya * rq : 0.0;
      const double bop = inb ? (q >= rb ? yu : 0.0) : ya;
      v4f64 acc = (v4f64){g[0], g[1], g[2], g[3]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
#pragma unroll
      for (int j = kb + 1; j < 4; ++j) {
        // entries right of the diagonal in rows below the block belong to folded
        // identity rows that are not active yet: they stay zero
        const int c = 4 * j + q;
        g[j] = (r > c0 + 3 && c > r) ? 0.0 : acc[j];
      }
    }
    BA_T16_STAMP(5 * kb + 4)
  }
  // scale column c by 1/sqrt(d_c) (v_rsq_f64 + Newton), diagonal = sqrt(d_c);
  // the folded inverse gets the same column scaling, its diagonal is 1/sqrt(d_c)
  dinv = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * j + q;
    const double d = __shfl(g[j], c + 16 * q, 64);  // pivot of this lane's column
    const bool ok = d > 1e-300;
    const double ds = ok ? d : 1.0;
    double y = __builtin_amdgcn_rsq(ds);
    const double hd = 0.5 * ds;
#pragma unroll
    for (int nr = 0; nr < 3; ++nr) y = y * fma(-hd * y, y, 1.5);
    double sq = ds * y;
    sq = fma(fma(-sq, sq, ds), 0.5 * y, sq);
    y = fma(fma(-sq, y, 1.0), y, y);
    g[j] = ok ? ((r == c) ? sq : g[j] * y) : 0.0;
    if (r == c) dinv = ok ? y : 0.0;
  }
  BA_T16_STAMP(20)
  return nbad;
}

// ---- the same factorisation with a SHORT dependent chain --------------------
// tile16_potrf_inv runs ~60 dependent fp64 instructions per 4-column block (four
// reciprocals with their Newton steps and validity selects one after the other,
// cross-lane gathers, a y-chain), ~1 800 cycles per block for a single wave.  Here:
//  * the 4x4 diagonal block is eliminated FRACTION-FREE (Bareiss): the leading minors
//    M1..M4 need no division on the way — f = (M2 e - e e^T) / M1 — and every
//    reciprocal is the square of a reciprocal square root, so the only transcendental
//    chains are four v_rsq_f64 (+ two Newton steps), two of them off the critical path:
//      q_k = M_k^-1/2,  1/M_k = q_k^2,  l_ik = (minor) q_k^2,  1/p_k = M_k q_(k+1)^2,
//      p_k^-1/2 = M_k q_k q_(k+1);
//  * no cross-lane gather and no y-chain: Y = P Ltilde^-T is ONE MFMA whose B operand is
//    the lane's own panel entry and whose A operand is Ltilde^-1 spread over the
//    lanes r < 4 (column q of Ltilde^-1 by a per-lane recurrence on constants, row r by
//    one select); the folded identity rows of the block rows are a second MFMA on
//    0/1 operands;
//  * the pivot-validity selects leave the chain: the minors are checked once per
//    block (wave-uniform) and a tile that meets a non-positive pivot is redone by
//    tile16_potrf_inv, which keeps the zeroed-column semantics;
//  * the 1/sqrt(d) column scaling needs no final pass of its own: the block's
//    p_k^-1/2 are at hand.
// Branch-free selects.  Given `c ? expensive_a : expensive_b` the compiler sinks the
// arms into divergent branches (s_and_saveexec / s_cbranch_execz per select: dozens of
// basic blocks in this routine, each a pipeline drain for the single wave that runs
// it).  Pinning the candidates (an empty asm the optimiser must treat as their
// definition) leaves it v_cndmask and nothing else.
#define BA_T16_PIN(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ double selb(bool c, double a, double b) {
  BA_T16_PIN(a);
  BA_T16_PIN(b);
  return c ? a : b;
}
__device__ __forceinline__ double sel4b(int k, double a0, double a1, double a2, double a3) {
  BA_T16_PIN(a0);
  BA_T16_PIN(a1);
  BA_T16_PIN(a2);
  BA_T16_PIN(a3);
  const double lo = k == 0 ? a0 : a1, hi = k == 2 ? a2 : a3;
  return k < 2 ? lo : hi;
}
// m^-1/2: v_rsq_f64 (about 2^-26) + one third-order step, e = 1 - m y^2,
// y (1 + e/2 + 3 e^2 / 8): five dependent operations to full precision
__device__ __forceinline__ double rsq_refined(double m) {
  const double y = __builtin_amdgcn_rsq(m);
  const double e = fma(-(m * y), y, 1.0);
  return fma(y * e, fma(0.375, e, 0.5), y);
}
// One 4-column block step of the short-chain factorisation: from the diagonal tile's
// register gk (= G[r][c0 + q]) the A operand of the Y products (Ltilde^-1 on the lanes
// r < 4), the reciprocal pivot rq and p^-1/2 (sck) of this lane's column c0 + q; `bad`
// collects non-positive minors (wave-uniform).
struct BlockStep {
  double aop1, rq, sck;
};
__device__ __forceinline__ BlockStep tile16_block_step(const double gk, const int c0, const int r, const int q,
                                                       const double b0, const double b1, const double b2,
                                                       const double b3, bool &bad) {
  // diagonal block, lower part: D[i][k] lives in lane (c0 + i, k)
  const double d00 = readlane_f64(gk, c0 + 0);
  const double w10 = readlane_f64(gk, c0 + 1), d11 = readlane_f64(gk, c0 + 1 + 16);
  const double w20 = readlane_f64(gk, c0 + 2), d21 = readlane_f64(gk, c0 + 2 + 16),
               d22 = readlane_f64(gk, c0 + 2 + 32);
  const double w30 = readlane_f64(gk, c0 + 3), d31 = readlane_f64(gk, c0 + 3 + 16),
               d32 = readlane_f64(gk, c0 + 3 + 32), d33 = readlane_f64(gk, c0 + 3 + 48);
  // Bareiss: e = M1 S1, f = M2 S2, h = M3 S3 (S_k: Schur complement after k pivots)
  const double M1 = d00;
  const double q1 = rsq_refined(M1), i1 = q1 * q1;
  const double e11 = fma(M1, d11, -(w10 * w10)), e21 = fma(M1, d21, -(w20 * w10)),
               e31 = fma(M1, d31, -(w30 * w10)), e22 = fma(M1, d22, -(w20 * w20)),
               e32 = fma(M1, d32, -(w30 * w20)), e33 = fma(M1, d33, -(w30 * w30));
  const double M2 = e11;
  const double q2 = rsq_refined(M2), i2 = q2 * q2;
  const double f22 = fma(M2, e22, -(e21 * e21)) * i1, f32 = fma(M2, e32, -(e31 * e21)) * i1,
               f33 = fma(M2, e33, -(e31 * e31)) * i1;
  const double M3 = f22;
  const double q3 = rsq_refined(M3), i3 = q3 * q3;
  const double M4 = fma(M3, f33, -(f32 * f32)) * i2;
  const double q4 = rsq_refined(M4), i4 = q4 * q4;
  // the pivots p_k = M_(k+1) / M_k are positive iff the leading minors are: all must
  // be safely positive (NaN fails too; a product of four tiny pivots that underflows
  // takes the slow path as well, which is merely slower)
  bad |= !((M1 > 1e-280) & (M2 > 1e-280) & (M3 > 1e-280) & (M4 > 1e-280));
  const double l10 = w10 * i1, l20 = w20 * i1, l30 = w30 * i1, l21 = e21 * i2, l31 = e31 * i2,
               l32 = f32 * i3;
  // column q of Ltilde^-1 (rows 0..3), then this lane's row r (lanes r < 4 feed the MFMA)
  const double x0 = b0;
  const double x1 = fma(-l10, x0, b1);
  const double x2 = fma(-l21, x1, fma(-l20, x0, b2));
  const double x3 = fma(-l32, x2, fma(-l31, x1, fma(-l30, x0, b3)));
  BlockStep o;
  o.aop1 = selb(r < 4, sel4b(r, x0, x1, x2, x3), 0.0);
  // reciprocal pivots and their square roots for this lane's column c0 + q
  o.rq = sel4b(q, i1, M1 * i2, M2 * i3, M3 * i4);
  o.sck = sel4b(q, q1, M1 * q1 * q2, M2 * q2 * q3, M3 * q3 * q4);
  return o;
}

__device__ __forceinline__ int tile16_potrf_inv2(double g[4], int lane, double &dinv) {
  const int r = lane & 15, q = lane >> 4;
  const double g_in[4] = {g[0], g[1], g[2], g[3]};
  // per-lane constants of the A-operand recurrence: column q of a unit lower 4x4 inverse
  const double b0 = q == 0 ? 1.0 : 0.0, b1 = q == 1 ? 1.0 : 0.0, b2 = q == 2 ? 1.0 : 0.0,
               b3 = q == 3 ? 1.0 : 0.0;
  double sc[4];  // p^-1/2 of this lane's column of every block
  bool bad = false;
  const v4f64 zero4 = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int c0 = 4 * kb;
    const BlockStep bs = tile16_block_step(g[kb], c0, r, q, b0, b1, b2, b3, bad);
    const int rb = r - c0;  // 0..3 inside the block
    const bool inb = rb >= 0 && rb < 4;
    const double idr = selb(inb & (rb == q), 1.0, 0.0);
    // Y[r][q] = sum_k P[r][k] Ltilde^-T[k][q]  (rows below the block: L sqrt(d); rows above
    // it: the folded inverse, same transform);  yu = row rb of Ltilde^-T for the block rows
    const double ya = __builtin_amdgcn_mfma_f64_16x16x4f64(bs.aop1, g[kb], zero4, 0, 0, 0)[0];
    const double yu = __builtin_amdgcn_mfma_f64_16x16x4f64(bs.aop1, idr, zero4, 0, 0, 0)[0];
    sc[kb] = bs.sck;
    g[kb] = selb(inb & (q > rb), yu, ya);
    if (kb < 3) {
      const double aop = selb(r > c0 + 3, -ya * bs.rq, 0.0);
      const double bop = selb(inb, selb(q >= rb, yu, 0.0), ya);
      v4f64 acc = (v4f64){g[0], g[1], g[2], g[3]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
#pragma unroll
      for (int j = kb + 1; j < 4; ++j) {
        const int c = 4 * j + q;
        g[j] = selb((r > c0 + 3) & (c > r), 0.0, acc[j]);
      }
    }
  }
  if (bad) {  // wave-uniform: the minors are the same in every lane
    g[0] = g_in[0]; g[1] = g_in[1]; g[2] = g_in[2]; g[3] = g_in[3];
    return tile16_potrf_inv(g, lane, dinv);
  }
  // column scaling: L = Y p^-1/2, diagonal = p^1/2 = Y_cc p^-1/2 as well (Y_cc = p_c);
  // the folded inverse gets the same scaling, its diagonal is p^-1/2
  dinv = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    g[j] *= sc[j];
    dinv = selb(r == 4 * j + q, sc[j], dinv);
  }
  return 0;
}

// ---- a 32x32 SPD tile by ONE wave: three 16x16 register tiles ---------------------
// g00 / g11: diagonal tiles in the folded layout above (lower = A, strict upper = 0 ->
// L^-T of THAT 16x16 tile); g10: the full off-diagonal tile, lane (r, q) reg j =
// A[16 + r][4 j + q].  Eight block steps of the short chain in a row: the first four
// also carry g10 (its Y product is a third MFMA with the same A operand) and update all
// three tiles (rank-4 MFMAs: the A operand of a product is indexed by the TARGET column,
// the B operand by the row); the last four are tile16_potrf_inv2's on g11.  Replaces, for
// 32-column tiles, factor_tile_lds' potrf -> barrier -> TRSM -> barrier -> SYRK -> barrier
// -> potrf: no barrier, no LDS round trip, one wave.  Returns non-zero (wave-uniform)
// when a minor is not safely positive: the tiles are then untouched and the caller takes
// the first path, which keeps the zeroed-column semantics.
__device__ __forceinline__ int tile32_potrf_inv(double g00[4], double g10[4], double g11[4], int lane,
                                                double &dinv0, double &dinv1) {
  const int r = lane & 15, q = lane >> 4;
  const double in00[4] = {g00[0], g00[1], g00[2], g00[3]}, in10[4] = {g10[0], g10[1], g10[2], g10[3]},
               in11[4] = {g11[0], g11[1], g11[2], g11[3]};
  const double b0 = q == 0 ? 1.0 : 0.0, b1 = q == 1 ? 1.0 : 0.0, b2 = q == 2 ? 1.0 : 0.0,
               b3 = q == 3 ? 1.0 : 0.0;
  double sc0[4], sc1[4];
  bool bad = false;
  const v4f64 zero4 = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {  // columns 0 .. 15
    const int c0 = 4 * kb;
    const BlockStep bs = tile16_block_step(g00[kb], c0, r, q, b0, b1, b2, b3, bad);
    const int rb = r - c0;
    const bool inb = rb >= 0 && rb < 4;
    const double idr = selb(inb & (rb == q), 1.0, 0.0);
    const double ya = __builtin_amdgcn_mfma_f64_16x16x4f64(bs.aop1, g00[kb], zero4, 0, 0, 0)[0];
    const double yu = __builtin_amdgcn_mfma_f64_16x16x4f64(bs.aop1, idr, zero4, 0, 0, 0)[0];
    const double y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bs.aop1, g10[kb], zero4, 0, 0, 0)[0];
    sc0[kb] = bs.sck;
    g00[kb] = selb(inb & (q > rb), yu, ya);
    g10[kb] = y1;
    // trailing columns of the first tile column (targets in g00 and g10): A operand by
    // target column = the rows of g00 below the block
    if (kb < 3) {
      const double aop = selb(r > c0 + 3, -ya * bs.rq, 0.0);
      const double bop = selb(inb, selb(q >= rb, yu, 0.0), ya);
      v4f64 acc = (v4f64){g00[0], g00[1], g00[2], g00[3]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
      v4f64 acc1 = (v4f64){g10[0], g10[1], g10[2], g10[3]};
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, y1, acc1, 0, 0, 0);
#pragma unroll
      for (int j = kb + 1; j < 4; ++j) {
        const int c = 4 * j + q;
        g00[j] = selb((r > c0 + 3) & (c > r), 0.0, acc[j]);
        g10[j] = acc1[j];
      }
    }
    // the second diagonal tile: target column = a row of g10 (all of them lie below the block)
    {
      v4f64 acc2 = (v4f64){g11[0], g11[1], g11[2], g11[3]};
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-y1 * bs.rq, y1, acc2, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) g11[j] = selb(4 * j + q > r, 0.0, acc2[j]);  // (its folded identity rows are not active yet)
    }
  }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {  // columns 16 .. 31: the second diagonal tile alone
    const int c0 = 4 * kb;
    const BlockStep bs = tile16_block_step(g11[kb], c0, r, q, b0, b1, b2, b3, bad);
    const int rb = r - c0;
    const bool inb = rb >= 0 && rb < 4;
    const double idr = selb(inb & (rb == q), 1.0, 0.0);
    const double ya = __builtin_amdgcn_mfma_f64_16x16x4f64(bs.aop1, g11[kb], zero4, 0, 0, 0)[0];
    const double yu = __builtin_amdgcn_mfma_f64_16x16x4f64(bs.aop1, idr, zero4, 0, 0, 0)[0];
    sc1[kb] = bs.sck;
    g11[kb] = selb(inb & (q > rb), yu, ya);
    if (kb < 3) {
      const double aop = selb(r > c0 + 3, -ya * bs.rq, 0.0);
      const double bop = selb(inb, selb(q >= rb, yu, 0.0), ya);
      v4f64 acc = (v4f64){g11[0], g11[1], g11[2], g11[3]};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
#pragma unroll
      for (int j = kb + 1; j < 4; ++j) {
        const int c = 4 * j + q;
        g11[j] = selb((r > c0 + 3) & (c > r), 0.0, acc[j]);
      }
    }
  }
  if (bad) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g00[j] = in00[j];
      g10[j] = in10[j];
      g11[j] = in11[j];
    }
    return 1;
  }
  dinv0 = 0.0;
  dinv1 = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    g00[j] *= sc0[j];
    g10[j] *= sc0[j];
    g11[j] *= sc1[j];
    dinv0 = selb(r == 4 * j + q, sc0[j], dinv0);
    dinv1 = selb(r == 4 * j + q, sc1[j], dinv1);
  }
  return 0;
}

}  // namespace tile16
}  // namespace ba
#endif  // BA_TILE16_H_
