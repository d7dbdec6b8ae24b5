// ba_device.h — device-side structures and launcher prototypes shared by the
// HIP translation units (internal; the public boundary is include/ba_hip.h).
#ifndef BA_DEVICE_H_
#define BA_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <vector>

namespace ba {

// LM controller state, resident in device memory so that a whole batch of
// iterations can be enqueued without host round trips.  Mirrors the locals of
// reference core/full_bundle_adjustment_solver.cpp:705-1008.
struct DevCtrl {
  double lambda;
  double prev_cost;
  double huber;
  double thr_step;
  double thr_cost;
  double dec_ratio;
  double inc_ratio;
  unsigned long long t_last;  // wall_clock64 stamp of the previous iteration
  int cur;        // which parameter buffer holds the accepted parameters
  int done;       // set once converged / max_iter reached: kernels no-op
  int iter;       // iterations completed
  int converged;
  int max_iter;
  int gn;         // 1: plain Gauss-Newton (always accept, lambda fixed)
  int lcur;       // which block buffer holds the linearisation at the accepted parameters
  // Parameter / block buffers of the TRIAL point of the iteration in flight,
  // written by k_pose_update before the trial-point linearisation starts.  The
  // pose side of that linearisation runs on the side stream and may execute after
  // the control step has flipped cur / lcur: it must not derive "trial" from them.
  int tcur, tlcur;
  int pad_;
};

struct DevIterRec {  // layout-identical to ba_iter_info
  double cost, cost_change, average_reprojection_error, abs_gradient, abs_step,
      damping_term, iter_time_ms;
  int iteration_status, pad_;
  double rho, model_change, trial_cost;
};

// All device pointers of one problem shard.
// B_ji = w Q^T R (6x3, reference :826) has the structure [K ; [X_ij]x K] with
// K = w G^T R (3x3, = rows 0..2 of B_ji bit for bit) and X_ij the point in the
// pose frame: Q = [G, G(-[X_ij]x)].  HBM holds the 12 doubles {K row-major,
// X_ij} per pair (96 B instead of 144); consumers rebuild rows 3..5 as
// X_ij x (column of K), or use B^T x = K^T (x_t + x_r x X_ij) directly.
constexpr int kWStride = 12;

struct DevProblem {
  // sizes
  int n_cam, n_pose, N, n_pt, M, M_global;
  int64_t n_obs, n_obs_opt, n_obs_global, P, n_pobs, T, B;
  int n_achunk, n_tchunk;
  // parameters: two buffers each (accepted / trial), selected by ctrl->cur
  double *cams;      // n_cam*16: fx fy cx cy R9 t3
  double *poses[2];  // n_pose*12
  double *pts[2];    // n_pt*3
  // landmark-major observations
  int4 *obs_idx;
  double2 *obs_uv;
  int64_t *lm_obs_ptr;
  int64_t *lm_pair_ptr;
  int32_t *pair_pose;
  int32_t *pair_lm;
  // pose-major observations
  int2 *pobs_idx;   // {camera, point}
  int2 *obs_cp;     // n_obs {camera | pose << 16, point} for k_cost, or null (>= 65536 cameras / poses)
  double2 *pobs_uv;
  int32_t *achunk_pose;
  int64_t *achunk_begin, *achunk_end;
  int32_t *pose_achunk_ptr;
  // pose-major pairs
  // Schur structure
  int32_t *sblk_j, *sblk_k;
  int32_t *diag_blk;
  int64_t *tri_p, *tri_q;
  int32_t *tchunk_blk;
  int64_t *tchunk_begin, *tchunk_end;
  int32_t *sblk_tchunk_ptr;
  // Schur super-runs (landmark-major, register-resident slot accumulators)
  int n_sup, n_slot;
  struct SupDesc {
    int32_t s0, ns, chunk_begin, chunk_end;
  };
  struct ChunkDesc {
    int64_t p0, tb, sp;
    int32_t l0, nl, np, nt;
  };
  // covisibility groups (k_schur_grp): layout-identical to Plan::GrpDesc
  struct GrpDesc {
    int64_t p0;
    int32_t l0, nl, d, s0;
    int32_t pose[20];
    int32_t pad_[6];
  };
  struct LinDesc {  // one k_lin_grp workgroup: layout-identical to Plan::LinDesc
    int64_t p0, o0;
    int32_t l0, nl, d, no;
    int32_t pat0;
    int32_t apart0, cost_idx;
    int32_t pad_;
  };
  LinDesc *lin_desc;
  int n_lin_desc;
  int n_lin_plain;  // pieces [0, n_lin_plain): exact groups; behind them: masked (superset) groups
  GrpDesc *grp32, *grp64, *grp128;  // pose sets of <= 5 / 6..10 / 11..20 poses
  int n_grp32, n_grp64, n_grp128;
  // k_lin_grp (groups linearised landmark and pose side in one pass): per pattern
  // slot {pose, camera | jj << 16 | optimisable << 29 | last writer << 30}
  int2 *grp_pat;
  double *Apart2;            // n_apart2 * 27 pose-side partial sums of the group pieces
  double *lin_dump;          // kLinDump doubles: target of k_lin_grp's lanes with nothing to store
  int *bl_flag, *pose_flag;  // k_backsub_lin: "piece back-substituted" (n_lin_desc) / "poses updated" (kPoseGrid) flags
  int32_t *pose_gpart_ptr, *pose_gpart;  // rows of Apart2 per pose
  int lin_chunk0;            // k_lin_landmarks starts at this chunk (the chunks before are k_lin_grp's)
  int n_lin_cost;            // entries of lin_cost_part: n_bchunk + k_lin_grp pieces
  int n_bs_grp;              // group-role workgroups of k_backsub_update (= n_lin_desc, or 0)
  int n_lm_part;             // entries of lm_part: n_bs_grp + chunk workgroups behind lin_chunk0
  SupDesc *sup_desc;
  uint32_t *sup_lane;  // n_sup*256: lane -> (slot, half, position, lanes per half) of k_schur_lds
  ChunkDesc *chunk_desc;
  uint16_t *chunk_sp;
  uint32_t *ltri;
  int64_t *blk_contrib_ptr;
  int32_t *contrib_slot;
  // the same per block as ONE 64-byte record (k_schur_final reaches its slot partials
  // after one dependent load): {j, k, first contribution, #contributions, first triple
  // chunk, #triple chunks, 0, 0, first eight slots}
  int32_t *blk_desc;
  double *spart2;  // n_slot*kSlotStride slot partial sums (36 of S + 6 of rhs)
  // Linearisation blocks: TWO buffers each, selected by ctrl->lcur.  Every LM
  // iteration linearises at its TRIAL point into the other buffer (the trial cost
  // is a by-product of that pass: no separate cost kernel); accepting the step
  // flips lcur together with cur, rejecting it keeps the old blocks, which are
  // still the linearisation at the accepted point (reference :943-953 only
  // changes lambda then).  The blocks are stored UNDAMPED; the (1 + lambda)
  // scaling of the diagonals (reference :833-852) is applied where they are read.
  double *Cu[2];   // M*6   C_i upper (00 01 02 11 12 22), undamped
  double *b[2];    // M*3
  double *W[2];    // P*kWStride  compact B_ji (see kWStride)
  double *A[2];    // N*36  full (mirrored), undamped
  double *a[2];    // N*6
  double *Cinv;    // M*6   inverse of the damped C_i (k_damp_invert, after the control step)
  double *Cd;      // M*6   damped C_i: written only for the readers (ba_get_C)
  double *Apart;   // n_achunk*27
  double *spart;   // n_tchunk*kSlotStride
  double *x;       // 6N
  double *y;       // M*3
  // scalar reductions
  double *cost_part;   // kCostGrid (k_cost: stage API, observations of fixed landmarks)
  double *lin_cost_part;  // n_bchunk: sum of residual norms per k_lin_landmarks workgroup
  int64_t n_obs_lm;    // observations of optimisable landmarks = the first n_obs_lm of the
                       // landmark-major list (k_lin_landmarks sees exactly these)
  double *lm_part;     // n_lm_part*2 : model (landmark side), sum |y| per landmark workgroup of k_backsub_update
  double *pose_part;   // [0..1] totals, then kPoseGrid*2 block partials:
                       // model (pose side), sum |x|
  double *scal;        // exchange buffer 1: [0] cost [1] model est [2] sum|y|
  // controller
  DevCtrl *ctrl;
  DevIterRec *log;
  int log_cap;
  // packed reduced system (exchange buffer 0): B*36 block entries, then 6N rhs
  double *Spk;
  // dense reduced system: column-major lower, ld rows (rhs rides as row npad)
  double *L;
  int npad, ld;
  double *Ldiag;   // (npad/nb) * dense_ws_per_block(nb) (diagonal factors + inverses)
  int nb;          // tile order of the reduced system (32 or 64)
  int *pose_col;   // N: first dense column of optimised pose j
  int *col_x;      // npad: dense column -> 6*pose + r, or -1 (padding)
  int32_t *bchunk_lm;  // n_bchunk+1 landmark ranges (backsub)
  // the same chunks as one 32-byte record each, so that a workgroup knows its
  // landmark / pair / observation ranges after ONE dependent load
  struct LmChunk {
    int64_t pb, ob;           // first pair, first (landmark-major) observation
    int32_t l0, nl, np, no;   // first landmark, #landmarks, #pairs, #observations
  };
  LmChunk *lm_chunk;   // n_bchunk
  int n_bchunk;
  int *zt_I, *zt_J;  // tiles of L that are (re)initialised every iteration
  int n_zt;
};

constexpr int kLinDump = 64 * 4 * 4;
// added to DenseDev::bad_pivots by a dataflow sweep whose bounded poll gave up
constexpr int kFlowTimeout = 1 << 20;
constexpr int kBsChunks = 4;     // chunks per chunk-role workgroup of k_backsub_update
constexpr int kCostGrid = 1792;  // 7 waves/SIMD resident on 256 CUs
constexpr int kLmGrid = 1024;
constexpr int kPoseGrid = 16;
constexpr int kSlotStride = 42;  // 6x6 block of B Cinv B^T + 6 of B Cinv b
// dense workspace per 64-column block: L11 (64x64) + four 16x16 tile inverses

// ---- optional per-kernel device timing (hipEvents around every launch) ----
// Enabled by ba_enable_stage_timing: bench.py uses it to measure the average
// launch duration of each kernel live, on the stream the kernels run on.
enum KernelId {
  K_COST = 0, K_LIN_LANDMARKS, K_LIN_POSES, K_POSE_FINALIZE, K_DENSE_INIT,
  K_SCHUR_LDS, K_SCHUR_PARTIAL, K_SCHUR_FINAL, K_SCATTER,
  K_CHOL_DIAG, K_CHOL_TRSM, K_CHOL_UPDATE, K_CHOL_BACK, K_CHOL_LEVEL, K_CHOL_DIAG_TRSM, K_CHOL_TAIL, K_BACKSUB_UPDATE,
  K_POSE_UPDATE, K_SCALARS, K_CONTROL, K_DAMP_INVERT, K_SCHUR_GRP, K_LIN_GRP, K_BACKSUB_LIN, K_COUNT
};
struct KernelTimer {
  bool on = false;
  std::vector<hipEvent_t> pool;          // grows on demand
  std::vector<int> ids;                  // kernel id of span k (events 2k,2k+1)
  double ms[K_COUNT] = {0};
  long calls[K_COUNT] = {0};
  void begin(int id, hipStream_t s);
  void end(hipStream_t s);
  void collect();                        // after a stream sync
  void reset();
};
extern thread_local KernelTimer *g_ktimer;
// launch a kernel, bracketed by events when a timer is installed
#define BA_LAUNCH(id, kernel, grid, block, stream, ...)                     \
  do {                                                                      \
    if (::ba::g_ktimer && ::ba::g_ktimer->on) ::ba::g_ktimer->begin(id, stream); \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);        \
    if (::ba::g_ktimer && ::ba::g_ktimer->on) ::ba::g_ktimer->end(stream);  \
  } while (0)

// Pins a wave-uniform 32-bit value: everything it depends on (typically a
// scalar load) has to be issued AND complete before this point, and the
// compiler cannot sink those loads below a later branch.  Used to request a
// workgroup's record together with the control word instead of one after the
// other (`if (ctrl->done) return;` first costs a full scalar-load latency).
#define BA_KEEP_S(x) asm volatile("" ::"s"(x))

// ---- launchers (ba_kernels.hip) ----
// sel: 0 = accepted parameters (block buffer lcur), 1 = trial parameters (the
// other block buffer)
// k_cost over the observations [begin, n_obs) of the landmark-major list
void launch_cost(const DevProblem &d, int sel, int64_t begin, hipStream_t s);
// pose side (A_j, a_j) and landmark side (C_i, b_i, W_ji + the cost partials) of
// the linearisation at the `sel` parameters
void launch_lin_poses(const DevProblem &d, int sel, hipStream_t s);
void launch_lin_landmarks(const DevProblem &d, int sel, hipStream_t s);
void launch_damp_invert(const DevProblem &d, hipStream_t s);
// dense_init + Schur accumulation + finalisation; direct: single GPU, S and rhs are
// also placed in the dense matrix (no k_scatter); with_init: also reset the factor tiles
void launch_schur(const DevProblem &d, bool direct, bool with_init, hipStream_t s);
void launch_schur_accumulate(const DevProblem &d, hipStream_t s);
void launch_schur_final(const DevProblem &d, bool direct, hipStream_t s);
void launch_backsub_update(const DevProblem &d, hipStream_t s, bool zero_tiles = false);
bool can_fuse_backsub_lin(const DevProblem &d);
void launch_backsub_lin(const DevProblem &d, hipStream_t s, bool zero_tiles, int gen, int *bad);
void launch_scatter(const DevProblem &d, hipStream_t s);
// cost_src: 0 = the k_cost partials only (stage API), 1 = the k_lin_landmarks
// partials (+ the k_cost partials of fixed-landmark observations, if any)
void launch_scalars_cost_only(const DevProblem &d, int cost_src, hipStream_t s);
void launch_scalars(const DevProblem &d, int cost_src, hipStream_t s);
void launch_scalars_and_control(const DevProblem &d, int cost_src, hipStream_t s,
                                int finalize_sel = -1);  // single GPU; finalize_sel >= 0: + k_pose_finalize workgroups
void launch_control(const DevProblem &d, hipStream_t s);
void launch_init_ctrl_cost(const DevProblem &d, hipStream_t s);

// ---- dense solver (ba_dense.hip) ----
// Factor the npad x npad lower matrix in d.L (with the rhs carried as row
// `npad`) and write x (6N entries, pose order).
// Device copies of the static work lists of the level-scheduled Cholesky
// (host side: DenseSchedule in ba_dense_sched.h).
struct DenseDev {
  int *row_ptr = nullptr, *rows = nullptr;       // per position
  int *item_t = nullptr, *item_I = nullptr;      // TRSM items
  int *tgt_I = nullptr, *tgt_J = nullptr;        // update targets
  int *tgt_src_ptr = nullptr, *src_t = nullptr;  // their source panels
  int *tgt_desc = nullptr, *back_desc = nullptr; // 8-int inline records
  int *row_desc = nullptr;                       // 16-int records: row tiles of a position
  // fused level path (DenseSchedule::fused_ok)
  int *f_desc = nullptr, *f_pend = nullptr;
  double *cbuf = nullptr;  // n_contrib contribution tiles (NB x NB, column-major)
  int *col_x = nullptr;   // npad: column -> index into x (6*pose + r) or -1
  double *xc = nullptr;   // npad: solution in column order
  int *bad_pivots = nullptr;  // device counter of non-positive pivots (or null)
  // dataflow backward sweep (k_chol_back_flow): tile positions top level first, one flag
  // per tile, the ticket counter; flow_gen = generation number of the next solve (host)
  int *flow_order = nullptr, *flow_flags = nullptr, *flow_ticket = nullptr;
  // forward sweep, one dataflow launch per level (k_chol_level_flow): one flag per tile
  // ("factorised, row tiles solved"), the ticket counter
  int *fwd_flags = nullptr, *fwd_ticket = nullptr;
  // forward sweep, ALL non-tail levels as one dataflow launch (k_chol_fwd_flow): the items
  // {kind, index} in level order, per update target the updates of earlier levels on its
  // column, per position all updates on its column, the columns' counters of finished updates
  int *fwd_items = nullptr, *upd_pre = nullptr, *col_need = nullptr, *fwd_cnt = nullptr;
  int n_fwd_items = 0, n_fwd_cnt = 0;
  // OPT-IN (BA_DENSE_FWD_FLOW=1; default: one dataflow launch per level).  Measured on MI355X:
  // the launch itself is shorter than the six it replaces (C4 104 vs 107 us, C2 112 vs 122,
  // C3 80 vs 87 under per-kernel timing) but the free-running LM iteration is SLOWER (C4
  // 0.438-0.446 vs 0.433 ms, C2 0.269 vs 0.266, C3 0.258 vs 0.2545): back-to-back launches
  // on one stream cost less than the polling workgroups of the later levels, which hold
  // their CU slots from the start of the launch
  bool want_fwd_flow = false;
  // forward sweep of the three-kernel path as one dataflow launch with lookahead (k_chol_dag):
  // items {kind, index}, TRSM items per position, "tile factorised" flags, per-column
  // counters of finished TRSM items (shares upd_pre / col_need / fwd_cnt / fwd_flags / fwd_ticket)
  int *dag_items = nullptr, *dag_ntrsm = nullptr, *dag_dflags = nullptr, *dag_tcnt = nullptr;
  int *look_need = nullptr;  // k_chol_look: per position, the previous level's targets in its column
  int n_dag_items = 0;
  // BA_DENSE_DAG=0: three launches per level; =1: also beyond kDagMaxItems.  Measured (MI355X):
  // moderately filled patterns gain — C1 0.304 -> 0.286 ms per iteration (forward sweep 188 ->
  // 145 us), W20 367 -> 303 us —, DENSE patterns LOSE: DENSE1K 5.5 -> 7.1 ms, n = 5 970 5.1 ->
  // 6.2 ms: 145 k update workgroups of ~3 us each pay ticket + descriptor + poll + late
  // target load (~5 us of dependent latency) at 2-3 workgroups per CU (the launch carries
  // the tile kernel's registers and LDS) against 4+ for the plain update kernel
  bool want_look2 = true, force_look2 = false;  // BA_DENSE_LOOK2=0: no in-launch lookahead on dense patterns; =1: instead of k_chol_dag too
  bool want_dag = true, force_dag = false;
  static constexpr int kDagMaxItems = 16384;
  int n_flow = 0, flow_tail_t0 = 0;
  mutable int flow_gen = 0;
  bool flow_ok = true;  // false while a hipGraph is captured / replayed (the generation is a kernel argument)
  bool want_flow = true;
  bool force_ticket = false;  // BA_DENSE_TICKET=1: tickets even when the grid is resident (test knob)
  // lookahead of the three-kernel (dense-pattern) path: an auxiliary stream and its events
  // (owned by the handle).  OPT-IN (BA_DENSE_LOOKAHEAD=1): measured SLOWER on MI355X / ROCm
  // 7.2 — two cross-stream event hops per level cost more than the factorisation they
  // hide (n = 5970: 6.5 vs 5.0 ms; DENSE1K 8.1 vs 6.7 ms; C1 0.42 vs 0.31 ms)
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_m = nullptr, ev_x[2] = {nullptr, nullptr};
  bool want_look = false;
  // BA_DENSE_FUSED / BA_DENSE_SPLIT / BA_DENSE_TAIL as found when the schedule was uploaded
  bool want_fused = false, want_split = false, want_tail = true;
  void read_env() {
    const char *f = getenv("BA_DENSE_FUSED"), *s = getenv("BA_DENSE_SPLIT"), *t = getenv("BA_DENSE_TAIL");
    want_fused = f && f[0] == '1';
    want_split = s && s[0] == '1';
    want_tail = !(t && t[0] == '0');
    const char *fl = getenv("BA_DENSE_FLOW");
    want_flow = !(fl && fl[0] == '0');
    const char *ff = getenv("BA_DENSE_FWD_FLOW");
    want_fwd_flow = ff && ff[0] == '1';
    const char *dg = getenv("BA_DENSE_DAG");
    want_dag = !(dg && dg[0] == '0');
    force_dag = dg && dg[0] == '1';
    const char *l2 = getenv("BA_DENSE_LOOK2");
    want_look2 = !(l2 && l2[0] == '0');
    force_look2 = l2 && l2[0] == '1';
    const char *tk = getenv("BA_DENSE_TICKET");
    force_ticket = tk && tk[0] == '1';
    const char *la = getenv("BA_DENSE_LOOKAHEAD");
    want_look = la && la[0] == '1';
  }
};
struct DenseSchedule;
void launch_dense_solve(const DevProblem &d, const DenseSchedule &sc,
                        const DenseDev &dd, hipStream_t s);
// stand-alone form for ba_dense_spd_solve (x receives n_x entries via col_x)
void dense_factor_solve(double *L, int npad, int ld, double *Ldiag, double *x,
                        const int *done_flag, const DenseSchedule &sc,
                        const DenseDev &dd, hipStream_t s);
// positions of the dataflow backward sweep (top level first); returns the tail block's first position
int dense_flow_order(const DenseSchedule &sc, const DenseDev &dd, std::vector<int> &order);
// work list of the one-launch forward sweep (k_chol_fwd_flow): items = {kind, index} pairs in
// level order (kind 0: tile position, 1: update target), pre[tg] = updates of earlier levels on
// the target's column, need[p] = all updates on position p's column; false if the path does not apply
bool dense_fwd_items(const DenseSchedule &sc, const DenseDev &dd, std::vector<int> &items,
                     std::vector<int> &pre, std::vector<int> &need);
// the same for the three-kernel path (k_chol_dag): kinds 0 tile / 1 TRSM item / 2 update target in
// lookahead order, ntrsm[p] = TRSM items of position p
bool dense_dag_items(const DenseSchedule &sc, const DenseDev &dd, std::vector<int> &items,
                     std::vector<int> &pre, std::vector<int> &need, std::vector<int> &ntrsm,
                     std::vector<int> &look_need);
void launch_dense_init(double *L, int ld, const int *col_x, const int *zt_I,
                       const int *zt_J, int n_zt, int nb, const int *done_flag,
                       hipStream_t s);

// ---- pose-only (ba_pose_only.hip) ----
struct PoIter {
  float cost, cost_change, abs_step;
};
int pose_only_sync_ints();       // size of the grid-barrier words (ints)
int pose_only_partial_floats();  // size of the partial-sum exchange buffer (floats)
int pose_only_mono6_device(const float *dX3, const float *duv2, int n, float fx,
                           float fy, float cx, float cy, float *dT12,
                           uint8_t *dmask, float thr_huber, float thr_step,
                           float thr_cost, float thr_out, int max_it,
                           PoIter *d_iters, int cap, int *d_meta,
                           float *d_debug, int *d_gsync, float *d_partial, hipStream_t s);
int pose_only_stereo6_device(const float *dX3, const float *duvl2, const float *duvr2, int n,
                             float fx, float fy, float cx, float cy, const float *d_cam_r16,
                             float *dT12, uint8_t *dmask_l, uint8_t *dmask_r, float thr_huber,
                             float thr_step, float thr_cost, float thr_out, int max_it,
                             PoIter *d_iters, int cap, int *d_meta, float *d_debug,
                             int *d_gsync, float *d_partial, hipStream_t s);

}  // namespace ba
#endif
