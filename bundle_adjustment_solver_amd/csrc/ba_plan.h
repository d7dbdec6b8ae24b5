// ba_plan.h — host-side problem plan for the HIP bundle-adjustment path.
//
// Replaces the reference's FinalizeParameters / SetProblemSize / connectivity
// build (reference core/full_bundle_adjustment_solver.cpp:182-206, :243-308,
// :668-700).  Instead of pointer-keyed hash maps and dense N x M block grids
// it produces flat, sorted index arrays that the kernels stream:
//   * landmark-major observation list (C/b/W side),
//   * pose-major observation list (A/a side),
//   * (landmark, pose) pair CSR for the cross blocks W_ji,
//   * pose-major pair permutation (rhs side),
//   * (j,k)-sorted triple list for the Schur complement blocks.
#ifndef BA_PLAN_H_
#define BA_PLAN_H_

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace ba {

// std::vector whose resize() leaves trivially constructible elements uninitialised:
// the planner's big arrays (observation lists: 200 MB at BASELINE config C4) are
// written exactly once, by parallel loops — a zero fill in front of that would touch
// every page a first time on ONE thread.
// Blocks of 4 MiB and more are 2 MiB-aligned and marked for transparent huge pages
// (madvise): a 120 MB list then takes 60 page faults instead of 30 000 on its first touch.
void *plan_big_alloc(size_t bytes);
void plan_big_free(void *p, size_t bytes);
constexpr size_t kPlanBigBytes = (size_t)4 << 20;
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = default_init_allocator<U>;
  };
  using std::allocator<T>::allocator;
  T *allocate(size_t n) {
    if (n * sizeof(T) >= kPlanBigBytes) return (T *)plan_big_alloc(n * sizeof(T));
    return std::allocator<T>::allocate(n);
  }
  void deallocate(T *p, size_t n) {
    if (n * sizeof(T) >= kPlanBigBytes) return plan_big_free(p, n * sizeof(T));
    std::allocator<T>::deallocate(p, n);
  }
  template <class U, class... Args>
  void construct(U *p, Args &&...args) {
    if constexpr (sizeof...(Args) == 0)
      ::new ((void *)p) U;
    else
      ::new ((void *)p) U(std::forward<Args>(args)...);
  }
};
template <class T>
using pvec = std::vector<T, default_init_allocator<T>>;

struct PlanInput {
  int n_cam = 0;
  int n_pose = 0;
  const uint8_t *pose_fixed = nullptr;
  int n_pt = 0;
  const uint8_t *pt_fixed = nullptr;
  int64_t n_obs = 0;
  const int32_t *obs_cam = nullptr;
  const int32_t *obs_pose = nullptr;
  const int32_t *obs_pt = nullptr;
  const double *obs_uv = nullptr;
  int rank = 0;
  int world = 1;
};

// Work-item granularities (shared with the kernels).
// observations per A/a partial-sum item (one wave): chosen at finalize so that
// about kPoseWaveTarget waves cover all observations
constexpr int kPoseChunkMin = 256, kPoseChunkMax = 4096;
constexpr int kPoseWaveTarget = 8192;
constexpr int kTriChunk = 256;     // triples per Schur partial-sum item
constexpr int kSchurPairs = 128;      // pairs staged in LDS per chunk
constexpr int kSchurLandmarks = 64;  // landmarks per chunk
static_assert(kSchurPairs <= 256 && kSchurLandmarks <= 256, "triple words hold 8-bit local indices");
constexpr int kSchurTri = 1024;       // triples per chunk (LDS resident)
constexpr int kSchurSlots = 128;      // distinct blocks per super-run (>= 2 lanes each)
constexpr int kSchurSuperLandmarks = 384;  // landmarks per super-run (upper bound)
constexpr int kSchurSuperMin = 32;         // ... lower bound
constexpr int kSchurRunTarget = 768;       // super-runs aimed at (3 workgroups x 256 CUs)
constexpr int kSchurInterleave = 32;       // residue classes of the in-window landmark interleave
constexpr int kSchurSuperChunks = 64;      // chunks per super-run (descriptor table in LDS)
// Covisibility groups (k_schur_grp): landmarks seen by the IDENTICAL set of
// optimisable poses.  Their Schur contributions sum_i V_ji W_ki^T are one dense
// (6d x 3n) (3n x 6d) product per group, which runs on the fp64 matrix cores.
constexpr int kGrpMaxPoses = 20;    // pose-set sizes handled by the group kernels (k_schur_grp: <= 5: 32-wide image, <= 10: 64-wide, <= 20: 128-wide)
constexpr int kGrpMinLandmarks = 24;  // smaller groups stay on the super-run path
constexpr int kGrpMaxLandmarks = 1024; // landmarks per group workgroup (larger groups are split)
constexpr int kGrpWidePiece = 104;    // landmarks per k_schur_grp_wide workgroup (13 stages of 8)
constexpr int kGrpMaxObs = 32;        // observations per landmark of a group (one lane each in k_lin_grp)
constexpr int kLinGrpSteps = 24;      // wave steps per k_lin_grp workgroup (landmarks: 4 waves x nlw x steps)
inline int lin_grp_nlw(int no) { return 64 / no < 7 ? 64 / no : 7; }  // landmarks per wave step

struct Plan {
  // ---- sizes ----
  int n_cam = 0;
  int n_pose = 0;        // all poses (replicated on every shard)
  int N = 0;             // optimised poses
  int n_pt_global = 0;   // all points of the full problem
  int M_global = 0;      // optimised points of the full problem
  int n_pt = 0;          // points owned by this shard (opt first, then fixed)
  int M = 0;             // optimised points owned by this shard
  int64_t n_obs_global = 0;
  int64_t n_obs = 0;     // observations of owned points
  int64_t n_obs_opt = 0; // ... whose point is optimisable (prefix of list)
  int64_t P = 0;         // (landmark, pose) pairs, both optimisable
  int64_t n_pobs = 0;    // observations whose pose is optimisable
  int64_t T = 0;         // Schur triples
  int64_t B = 0;         // non-zero upper blocks of S (incl. all diagonals)

  // ---- index maps ----
  std::vector<int32_t> pose_int_of_user, pose_user_of_int;  // n_pose
  std::vector<int32_t> jopt_of_user;   // user pose -> opt index (input order)
  std::vector<int32_t> pt_int_of_user; // n_pt_global, -1 if not owned
  std::vector<int32_t> pt_user_of_int; // n_pt
  std::vector<int32_t> iopt_of_user;   // user point -> GLOBAL opt idx or -1
  std::vector<int32_t> owner;          // n_pt_global owner rank

  // ---- landmark-major observations ----
  pvec<int32_t> obs_idx;          // n_obs*4: cam, pose_int, pt_int, pair|-1
  pvec<int32_t> obs_cp;           // n_obs*2: cam | pose_int << 16, pt_int (k_cost's slim record; empty with >= 65536 cameras / poses)
  pvec<double> obs_uv;            // n_obs*2
  std::vector<int64_t> lm_obs_ptr;   // M+1
  // ---- pairs ----
  std::vector<int64_t> lm_pair_ptr;  // M+1
  pvec<int32_t> pair_pose;           // P (internal pose, < N)
  pvec<int32_t> pair_lm;             // P (internal point, < M)
  // ---- pose-major observations (optimisable poses only) ----
  std::vector<int32_t> pobs_idx;  // n_pobs*2: cam, pt_int (pose-major copy: the pose is implied)
  std::vector<double> pobs_uv;    // n_pobs*2
  std::vector<int64_t> pose_obs_ptr;     // N+1
  std::vector<int32_t> achunk_pose;      // per A/a work item
  std::vector<int64_t> achunk_begin;     // nchunk+1 style: begin of each
  std::vector<int64_t> achunk_end;
  std::vector<int32_t> pose_achunk_ptr;  // N+1
  // ---- pose-major pair permutation ----
  // ---- Schur triples ----
  std::vector<int32_t> sblk_j, sblk_k;   // B
  std::vector<int32_t> diag_blk;         // N: block id of (j,j)
  std::vector<int64_t> sblk_tri_ptr;     // B+1
  std::vector<int64_t> tri_p, tri_q;     // T : pair ids (pose j side, k side)
  std::vector<int32_t> tchunk_blk;       // per Schur work item
  std::vector<int64_t> tchunk_begin, tchunk_end;
  std::vector<int32_t> sblk_tchunk_ptr;  // B+1
  // (the global triple list above holds only "big" landmarks; everything
  //  else goes through the super-runs below)
  struct SupDesc {                       // one Schur workgroup
    int32_t s0, ns;                      // its slots [s0, s0+ns)
    int32_t chunk_begin, chunk_end;      // its chunks
  };
  struct ChunkDesc {                     // <= kSchurPairs pairs staged at once
    int64_t p0, tb, sp;                  // first pair, first triple, chunk_sp base
    int32_t l0, nl, np, nt;              // landmarks, pairs, triples
  };
  std::vector<SupDesc> sup_desc;
  std::vector<uint32_t> sup_lane;        // 256 lane words per super-run (deal_lanes)
  std::vector<ChunkDesc> chunk_desc;
  std::vector<uint16_t> chunk_sp;        // per chunk: ns+1 slot offsets
  std::vector<int32_t> slot_blk;         // block of each slot
  std::vector<uint32_t> ltri;            // (local pair p << 16) | (local landmark << 8) | local pair q
  std::vector<int64_t> blk_contrib_ptr;  // B+1 -> contrib_slot
  std::vector<int32_t> contrib_slot;     // slots of each block, workgroup order
  std::vector<int32_t> bchunk_lm;        // landmark ranges of the backsub chunks
  // ---- covisibility groups: the first M_grp landmarks, ordered by group ----
  // A group's landmarks have the identical OBSERVATION PATTERN: the same sequence of
  // (pose, camera) over their observations in landmark-major order (fixed poses
  // included), hence the identical set of d optimisable poses.
  int M_grp = 0;
  bool lin_groups = false;               // the groups are also linearised by k_lin_grp
  // landmarks [l0, l0 + nl): d free poses, no pattern slots each.  masked: the pattern
  // is the UNION of the members' patterns (grp_upat[upat0 .. upat0 + no): (pose << 32) |
  // camera); a member's missing observations are padded slots with uv = NaN, its
  // missing pairs zero W records
  struct GrpRange {
    int32_t l0, nl, d, no;
    int32_t masked;
    int64_t upat0;
  };
  std::vector<uint64_t> grp_upat;
  std::vector<uint8_t> pair_pad;         // P: 1 = padded pair of a masked group (no observation)
  int64_t n_pair_pad = 0;
  std::vector<GrpRange> grp_range;
  struct GrpDesc {                       // one k_schur_grp workgroup (128 bytes)
    int64_t p0;                          // first pair: pair(il, jj) = p0 + d * il + jj
    int32_t l0, nl, d, s0;               // landmarks, pose count, first of its d (d + 1) / 2 slots
    int32_t pose[kGrpMaxPoses];          // ascending optimised pose indices
    int32_t pad_[6];
  };
  struct LinDesc {                       // one k_lin_grp workgroup (48 bytes): a run of landmarks of one group
    int64_t p0, o0;                      // pair(il, jj) = p0 + d * il + jj;  obs(il, oo) = o0 + no * il + oo
    int32_t l0, nl, d, no;               // landmarks, free poses, observations per landmark
    int32_t pat0;                        // first entry of the group's pattern in grp_pat
    int32_t apart0, cost_idx;            // first of its d rows of Apart2, its entry of lin_cost_part
    int32_t pad_;                        // 1 = piece of a masked (superset) group
  };
  std::vector<LinDesc> lin_desc;         // plain pieces [0, n_lin_plain), then the masked ones (pad_ = 1)
  int n_lin_plain = 0;
  std::vector<GrpDesc> grp32, grp64, grp128;  // d <= 5 / 6 <= d <= 10 / 11 <= d <= 20
  // pattern entry of observation slot oo: {pose (internal index), camera | jj << 16 |
  // optimisable pose << 29 | last writer of its pair << 30}
  std::vector<int32_t> grp_pat;          // 2 ints per slot
  int64_t n_apart2 = 0;                  // rows (27 doubles) of Apart2 = sum of d over the k_lin_grp pieces
  std::vector<int32_t> pose_gpart_ptr, pose_gpart;  // N+1 / rows of Apart2 that belong to pose j
  int n_bchunk_grp = 0;                  // back-substitution chunks that cover the grouped landmarks
};

// Tile pattern of the GLOBAL reduced camera matrix (all shards) for tiles of
// `poses_per_tile` consecutive optimised poses: ncb tiles, ncb*ncb symmetric
// adjacency bytes.
void tile_pattern(const Plan &pl, int poses_per_tile, int &ncb,
                  std::vector<uint8_t> &adj);

// Owner rank of every point: locality order (first observing optimised pose,
// then input index), contiguous chunks balanced by observation count.
void partition_points(const PlanInput &in, std::vector<int32_t> &owner);

// Returns empty string on success, otherwise an error message.
std::string build_plan(const PlanInput &in, Plan &plan);

}  // namespace ba
#endif
