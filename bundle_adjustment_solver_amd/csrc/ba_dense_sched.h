// ba_dense_sched.h — host-side symbolic analysis for the structure-aware,
// level-scheduled Cholesky of the reduced camera system (ba_dense.hip).
//
// The reference solves the reduced camera system with a dense, sequential,
// unblocked LDLT (reference core/full_bundle_adjustment_solver.cpp:905).  The
// matrix is block sparse: two poses couple only through landmarks they both
// observe.  Here the 6x6 pose blocks are grouped into 64-column TILES
// (kPosesPerTile poses + padding); on the tile graph we compute
//   1. a parallel minimum-degree ordering: each LEVEL is an independent set of
//      low-degree tiles (for a band matrix this is odd-even cyclic reduction,
//      log2(n) levels; for a dense matrix it degenerates to one tile per level,
//      i.e. the classic right-looking sweep);
//   2. the symbolic factor (fill-in) under that ordering;
//   3. static work lists per level: diagonal tiles, TRSM items, and update
//      TARGETS with their source panels (target-centric so that tiles shared by
//      several panels of one level are summed by one workgroup in fixed order:
//      deterministic, no atomics).
// Skipping structurally zero tiles is exact (they are zero in a dense
// factorisation too); only the floating-point summation order differs from the
// natural ordering.
#ifndef BA_DENSE_SCHED_H_
#define BA_DENSE_SCHED_H_

#include <cstdint>
#include <vector>

namespace ba {

constexpr int kMaxContrib = 32768;    // contribution tiles (8 KiB each at order 32)
// Tile order nb = 32 (5 poses + 2 padding columns) or 64 (10 poses + 4): chosen
// per problem by ba_finalize from the two level schedules.
inline int dense_poses_per_tile(int nb) { return nb == 32 ? 5 : 10; }
inline int dense_ws_per_block(int nb) { return nb * nb + (nb / 16) * 256; }
// row tiles a fused-level workgroup can keep in LDS (ba_dense_tile.inc: FR)
inline int dense_max_fused_rows(int nb) { return nb == 32 ? 12 : 3; }

struct DenseSchedule {
  int nb = 32;    // tile order the schedule was built for
  int ncb = 0;    // tiles (the rhs row block has index ncb)
  int nlev = 0;
  std::vector<int> pos_of_tile;  // original tile/group -> elimination position
  std::vector<int> tile_at_pos;  // inverse
  std::vector<int> lev_ptr;      // nlev+1: positions [lev_ptr[l], lev_ptr[l+1])
  std::vector<int> row_ptr, rows;  // per position: non-zero row tiles below
                                   // (ascending positions), rhs block last
  std::vector<int> item_ptr, item_t, item_I;        // TRSM items per level
  std::vector<int> tgt_ptr, tgt_I, tgt_J;           // update targets per level
  std::vector<int> tgt_first;                       // per level: its first targets that lie in a column of the NEXT level
  std::vector<int> tgt_src_ptr, src_t;              // sources of each target
  // The same lists as fixed 8-int records, so that a workgroup reaches its data
  // after ONE dependent load:
  //   tgt_desc[8 tg ..] = { I, J, nsrc, src_begin, first four sources (-1 pad) }
  //   back_desc[8 p ..] = { nrow (without the rhs block), row_begin, first six rows }
  std::vector<int> tgt_desc, back_desc;
  //   row_desc[16 p ..] = { nrow (incl. the rhs block), row_begin, 0.., first eight row tiles at [8..15] }
  std::vector<int> row_desc;
  int max_rows = 0;  // most row tiles (incl. the rhs block) below any tile
  // FUSED level schedule (one launch per level, see k_chol_level): every source
  // tile writes its outer products P_a P_c^T as separate CONTRIBUTION tiles
  // instead of updating the targets in place; a tile subtracts its pending
  // contributions when it is consumed (as a diagonal tile or as a row tile of
  // the column being eliminated).  Used when every tile has at most
  // dense_max_fused_rows(nb) row tiles (incl. the rhs block) and the number of
  // contribution tiles stays below kMaxContrib (banded / block-sparse systems);
  // dense patterns keep the in-place three-kernel path.
  //   f_desc[16 p ..] = { nrow, row_begin, pend_begin, pend_n, out_base, npairs, 0, 0,
  //                       first eight row tiles (-1 pad) }
  //   f_pend[2 k ..]  = { row slot a (-1: the diagonal tile), contribution id }:
  //                     the pending contributions of position p's tiles
  bool fused_ok = false;
  int n_contrib = 0;
  std::vector<int> f_desc, f_pend;
  double fill = 1.0;        // non-zero factor tiles / all lower tiles
  double flops = 0.0;       // executed flops of factor + solves (estimate)
};

// `adj` is the symmetric ncb x ncb tile adjacency (non-zero off-diagonal
// tiles), row-major bytes.  `natural_order` = keep the given order and put
// every tile in its own level (debug / dense comparison).
void build_dense_schedule(int ncb, const std::vector<uint8_t> &adj,
                          bool natural_order, int nb, DenseSchedule &s);

}  // namespace ba
#endif
