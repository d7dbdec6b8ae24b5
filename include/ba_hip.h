/*
 * ba_hip.h — C ABI of the MI355X (gfx950) bundle-adjustment hot path.
 *
 * The reference has no FFI / plugin layer: its boundary is the C++ class
 * FullBundleAdjustmentSolver (reference core/full_bundle_adjustment_solver.h:
 * 127-146) whose Solve() (reference core/full_bundle_adjustment_solver.cpp:
 * 630-1044) is the hot path.  This header is the thin C ABI a replacement
 * facade binds: plain pointers and sizes, an opaque handle,
 * no C++ or torch types.  Every entry point names the reference lines it
 * replaces.  Conventions:
 *   - return 0 = OK, negative = error (message via ba_last_error()); no C++
 *     exception crosses the ABI;
 *   - caller-owned HOST arrays are copied at set_* time;
 *   - all values are in the solver's SCALED units: the facade applies the
 *     reference's 0.01 scaling (reference :38, :74-79, :97, :113, :176),
 *     inverts user poses into T_jw (reference :96) and maps pointers to
 *     indices before calling in;
 *   - rigid transforms are 12 doubles: row-major 3x3 rotation, then t;
 *   - one handle per host thread; a handle owns one GPU.
 */
#ifndef BA_HIP_H_
#define BA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ba_handle ba_handle;

/* Mirrors Options (reference core/solver_option_and_summary.h:47-71); the
 * fields are float there and are promoted to double inside the LM update
 * (reference core/full_bundle_adjustment_solver.cpp:949,953). */
typedef struct {
  float threshold_step_size;         /* convergence_handle  */
  float threshold_cost_change;       /* convergence_handle  */
  float threshold_huber_loss;        /* outlier_handle      */
  float threshold_outlier_rejection; /* unused by full BA   */
  int max_num_iterations;            /* iteration_handle    */
  float initial_lambda;              /* trust_region_handle */
  float decrease_ratio_lambda;
  float increase_ratio_lambda;
  /* 1 = plain Gauss-Newton of FullBundleAdjustmentSolverRefactor (reference
   * core/full_bundle_adjustment_solver_refactor.cpp:976-982: every step
   * accepted, lambda fixed at initial_lambda); 0 = Levenberg-Marquardt, the
   * only mode of FullBundleAdjustmentSolver::Solve (its solver_type is
   * ignored, reference :630-1044). */
  int gauss_newton;
} ba_options;

/* One row per LM iteration: OptimizationInfo (reference
 * core/solver_option_and_summary.h:37-46) + the trust-region internals. */
typedef struct {
  double cost;
  double cost_change;
  double average_reprojection_error;
  double abs_gradient;
  double abs_step;
  double damping_term;
  double iter_time_ms;
  int iteration_status; /* 0 UPDATE, 1 UPDATE_TRUST_MORE, 2 SKIPPED */
  int pad_;
  double rho;
  double model_change;
  double trial_cost;
} ba_iter_info;

/* ---- lifetime --------------------------------------------------------- */
/* replaces the constructor / destructor / Reset (reference :6-70) */
int ba_create(ba_handle **out, int device_id);
void ba_destroy(ba_handle *h);
const char *ba_last_error(void);
/* Run all kernels of this handle on the given hipStream_t (used as is: NULL
 * is HIP's default stream).  Without this call the handle uses a stream of
 * its own. */
int ba_set_stream(ba_handle *h, void *hip_stream);

/* ---- problem construction (host arrays, copied) ------------------------ */
/* AddCamera, reference :72-85.  intr4 = fx,fy,cx,cy ; T_cj12 = body->camera */
int ba_set_cameras(ba_handle *h, int n_cam, const double *intr4,
                   const double *T_cj12);
/* AddPose + MakePoseFixed, reference :87-101, :119-134.  T_jw = pose^-1 */
int ba_set_poses(ba_handle *h, int n_pose, const double *T_jw12,
                 const uint8_t *fixed);
/* AddPoint + MakePointFixed, reference :103-117, :136-153 */
int ba_set_points(ba_handle *h, int n_pt, const double *X3,
                  const uint8_t *fixed);
/* AddObservation, reference :155-180.  Insertion order is preserved: it
 * decides which camera's cross term B_ji survives (reference :826). */
int ba_set_observations(ba_handle *h, int64_t n_obs, const int32_t *cam,
                        const int32_t *pose, const int32_t *point,
                        const double *uv2);
/* Landmark-range sharding for multi-GPU (new).  Call before
 * ba_finalize with the same full problem on every rank. */
int ba_set_shard(ba_handle *h, int rank, int world);
/* FinalizeParameters + SetProblemSize + connectivity, reference :182-206,
 * :243-308, :668-700: index assignment, block-sparse structure, upload. */
int ba_finalize(ba_handle *h);

/* New VALUES for the same STRUCTURE (a SLAM back end re-optimising the same graph):
 * replaces the parameters of a finalized problem — T_jw12 for all n_pose poses
 * and / or X3 for all n_pt points in user order, NULL = keep — without planning
 * again (index assignment, block structure, schedules and uploads of ba_finalize
 * stay: 0.2-0.6 s at BASELINE config C4).  Fixed flags, cameras and observations
 * cannot change.  The next ba_solve / ba_lm_begin starts from the new values.  (The
 * reference keeps its registered state across Solve calls, :44-70, and has no way
 * to re-seed it short of Reset.) */
int ba_update_values(ba_handle *h, const double *T_jw12, const double *X3);

/* Host-only helper (no GPU needed): owner rank of every point under the
 * sharding rule used by ba_finalize. */
int ba_partition_points(int n_pose, const uint8_t *pose_fixed, int n_pt,
                        const uint8_t *pt_fixed, int64_t n_obs,
                        const int32_t *obs_pose, const int32_t *obs_pt,
                        int world, int32_t *owner_out);

/* ---- multi-GPU exchange hook ------------------------------------------- */
/* The caller owns the collective (RCCL through torch.distributed, or
 * anything else).  `which`: 0 = reduced camera system, PACKED: the 36 entries
 * of every structurally non-zero 6x6 block of S (global block numbering,
 * identical on all shards) followed by the 6N entries of rhs; 1 = LM scalars.
 * The hook must sum-all-reduce n doubles at dev_ptr in place, ordered on
 * hip_stream. */
typedef int (*ba_allreduce_fn)(void *user, int which, void *dev_ptr,
                               int64_t n_doubles, void *hip_stream);
int ba_set_allreduce(ba_handle *h, ba_allreduce_fn fn, void *user);
/* `which` = 2 (ba_gather_points only): every point of the FULL problem in user
 * order, 3 doubles each, the rows of points this shard does not own zero. */
/* Size (in doubles) of exchange buffer `which` (0, 1 or 2), valid after ba_finalize. */
int64_t ba_reduce_buffer_size(ba_handle *h, int which);
/* Use caller-allocated DEVICE memory for exchange buffer `which` (so that a
 * framework tensor can alias it).  Call after ba_finalize. */
int ba_bind_reduce_buffer(ba_handle *h, int which, void *dev_ptr,
                          int64_t n_doubles);

/* Write-back under sharding.  The reference updates EVERY registered, non-fixed
 * point through the caller's pointer at the end of Solve (reference :1018-1022);
 * a shard holds the final values of the landmarks it owns only.  This call
 * sum-all-reduces the owned rows (hook, which = 2: 12 MB at BASELINE config C4,
 * once per Solve); afterwards, and until the next ba_lm_begin / ba_lm_iterate /
 * ba_stage_commit, ba_get_points returns every point of the full problem
 * (owned_mask all 1) on every rank.  A no-op without a hook or with world = 1. */
int ba_gather_points(ba_handle *h);

/* ---- RCCL exchange (csrc/ba_rccl.cpp) ----------------------------------- */
/* The hook above implemented over RCCL inside the library, so that no host
 * language runs between the kernels of an LM iteration: rank 0 draws the 128-byte
 * communicator id and hands it to the other ranks by any side channel (a file,
 * MPI, a torch.distributed broadcast); every rank creates its communicator and
 * registers  ba_set_allreduce(h, ba_rccl_allreduce_hook, comm).  librccl is
 * bound with dlopen at the first call (BA_RCCL_LIB, else librccl.so.1 — the copy
 * already in the process if a framework brought one). */
typedef struct ba_rccl_comm ba_rccl_comm;
int ba_rccl_available(void);                         /* 1 / 0 (reason: ba_last_error) */
int ba_rccl_get_unique_id(uint8_t id128[128]);       /* ncclGetUniqueId */
int ba_rccl_comm_create(ba_rccl_comm **out, int rank, int world,
                        const uint8_t id128[128], int device_id);
int ba_rccl_comm_size(ba_rccl_comm *c);              /* ranks the communicator sees (ncclCommCount) */
int64_t ba_rccl_comm_calls(ba_rccl_comm *c);         /* all-reduces issued so far */
void ba_rccl_comm_destroy(ba_rccl_comm *c);
/* a ba_allreduce_fn; user = ba_rccl_comm* */
int ba_rccl_allreduce_hook(void *user, int which, void *dev_ptr,
                           int64_t n_doubles, void *hip_stream);

/* ---- the LM loop ------------------------------------------------------- */
/* Solve, reference :630-1044 (iteration loop :705-1008).  Runs until
 * convergence or max_num_iterations; fills up to `cap` rows. */
int ba_solve(ba_handle *h, const ba_options *opt, ba_iter_info *out, int cap,
             int *n_iter, int *converged);
/* The same loop in three asynchronous pieces (bench / graph replay):
 * begin = lambda0 + initial cost (reference :707-708); iterate = enqueue n
 * LM iterations without host synchronisation (iterations after convergence
 * are device-side no-ops); sync = wait and read the iteration log. */
int ba_lm_begin(ba_handle *h, const ba_options *opt);
int ba_lm_iterate(ba_handle *h, int n);
/* returns 1 when the loop has finished, 0 when not, < 0 on error — also when a
 * dataflow hand-off of the reduced solve timed out on the device (the sweeps of
 * csrc/ba_dense_tile.inc poll with a bound; x is then partly unsolved and every
 * iteration since ba_lm_begin is invalid; ba_last_error names it). */
int ba_lm_sync(ba_handle *h, ba_iter_info *out, int cap, int *n_iter,
               int *converged);

/* ---- stage entry points (parity tests, per-stage timing) --------------- */
int ba_stage_cost(ba_handle *h, double *cost);            /* :381-433 */
int ba_stage_linearize(ba_handle *h, double lambda,
                       double huber);                     /* :716-856 */
int ba_stage_schur(ba_handle *h);                         /* :858-902 */
int ba_stage_solve_reduced(ba_handle *h);                 /* :905-908 */
int ba_stage_backsub_update(ba_handle *h);  /* :910-926, :435-455, :960-963 */
/* after ba_stage_backsub_update: trial cost, model change, step norms */
int ba_stage_scalars(ba_handle *h, double *trial_cost, double *model_change,
                     double *pose_step_sum, double *point_step_sum);
/* accept (1) or reject (0) the trial parameters, reference :939-945 */
int ba_stage_commit(ba_handle *h, int accept);

/* Accumulated device time per stage in ms since the last reset (hipEvents;
 * enabled by ba_enable_stage_timing): [0] build (linearize), [1] schur,
 * [2] reduced solve, [3] backsub+update, [4] cost, [5] control,
 * [6] exchange (all-reduce hook), [7] reserved. */
int ba_enable_stage_timing(ba_handle *h, int on);
int ba_get_stage_ms(ba_handle *h, double out8[8], int reset);

/* ---- readers (user index order; opt order = input order of non-fixed) -- */
int ba_num_opt_poses(ba_handle *h);
int ba_num_opt_points(ba_handle *h); /* local (owned) optimisable points */
int64_t ba_num_pairs(ba_handle *h);
int64_t ba_num_schur_blocks(ba_handle *h);
int64_t ba_num_schur_triples(ba_handle *h);
/* current parameters, write-back source (reference :1011-1022) */
int ba_get_poses(ba_handle *h, double *T_jw12 /* n_pose*12 */);
/* points not owned by this shard are left untouched; owned_mask may be NULL */
int ba_get_points(ba_handle *h, double *X3 /* n_pt*3 */, uint8_t *owned_mask);
/* damped A_j (6x6 full) and a_j, per optimised pose */
int ba_get_A(ba_handle *h, double *A36, double *a6);
/* damped C_i (3x3 full), b_i; Cinv_i, Cinv_i b_i.  Indexed by GLOBAL opt
 * point index; entries of points owned by other shards are left untouched */
int ba_get_C(ba_handle *h, double *C9, double *b3);
int ba_get_Cinv(ba_handle *h, double *Cinv9, double *Cinvb3);
/* pairs (global i_opt, j_opt) in internal order with W = B_ji (6x3, row-major; the
 * device keeps it compact as {K, X_ij} and this call expands it) */
int ba_get_pairs(ba_handle *h, int32_t *pair_i, int32_t *pair_j, double *W18);
/* reduced camera system, (6N)^2 row-major, and rhs, in opt-pose order */
int ba_get_S(ba_handle *h, double *S, double *rhs);
int ba_get_xy(ba_handle *h, double *x6, double *y3);

/* Per-kernel device time (hipEvents around every launch, recorded on the
 * handle's stream) accumulated while stage timing is enabled: total ms and
 * number of launches per kernel id in [0, ba_kernel_count()). */
int ba_kernel_count(void);
const char *ba_kernel_name(int id);
int ba_get_kernel_ms(ba_handle *h, double *ms_out, int64_t *calls_out, int reset);

/* Structure of the reduced-system factorisation chosen at ba_finalize:
 * out4 = { non-zero tile fraction of the factor (1 = dense), executed flop
 * estimate per solve, number of elimination levels (launch depth), padded
 * matrix order }. */
int ba_get_dense_info(ba_handle *h, double out4[4]);

/* How the Schur complement of this shard is accumulated (chosen at ba_finalize):
 * out8 = { workgroups of the covisibility-group kernel with 32-wide tiles (pose
 * sets of <= 5 poses), the same with 64-wide tiles (6..10 poses), landmarks
 * covered by groups, super-runs (the other landmarks), (landmark, pose) pairs in
 * groups, Schur triples in groups, v_mfma_f64_16x16x4 instructions the group
 * kernel executes per launch, triples on the global list (landmarks too large
 * for a super-run) }. */
int ba_get_schur_info(ba_handle *h, int64_t out8[8]);

/* How this shard is linearised and back-substituted (chosen at ba_finalize):
 * out4 = { workgroups (pieces) of the covisibility-group linearisation kernel —
 * landmark and pose side of the grouped landmarks in one pass; 0 if the groups
 * are linearised by the chunk / pose-major kernels —, observations those pieces
 * cover, landmark chunks left to the chunk kernel, observations on the
 * pose-major list }. */
int ba_get_lin_info(ba_handle *h, int64_t out4[4]);

/* Superset ("masked") covisibility groups — landmarks of one pose span whose
 * observation patterns differ (occlusion, image borders, track loss) share the UNION
 * of their patterns; a member's missing observations are padded slots of weight 0, its
 * missing pairs zero W records: out4 = { k_lin_grp pieces of masked groups, landmarks
 * in masked groups, padded observation slots, padded (landmark, pose) pairs }.  The
 * padding is internal: ba_num_pairs / ba_get_pairs do not show it. */
int ba_get_mask_info(ba_handle *h, int64_t out4[4]);

/* The reduced system is factorised by Cholesky WITHOUT pivoting; the
 * reference uses Eigen's diagonally pivoted LDLT with a pseudo-inverted D
 * (reference :905).  A non-positive pivot (<= 1e-300: a pose without
 * observations, or an indefinite / rank-deficient S, e.g. no fixed pose and
 * lambda at its floor) zeroes its column and solution component instead.
 * `count` receives how many such pivots the factorisations met since
 * ba_lm_begin (or since the last reset): 0 means the two factorisations agree
 * up to roundoff.  (Timed-out hand-offs of the dataflow sweeps are NOT counted
 * here: they are errors, reported by ba_lm_sync.) */
int ba_get_dropped_pivots(ba_handle *h, int64_t *count, int reset);

/* ---- observation streaming: problems larger than device memory ---------- */
/* SURVEY.md 8f N4 (the reference has no counterpart: its dense N x M grids stop it
 * at 62 GB of host memory long before).  The landmarks are cut into n_chunks
 * chunks by the rule of ba_set_shard; a chunk's observations, W = B_ji, C_i, b_i and
 * points live in ONE of two device arenas of arena_bytes each and in a pinned host
 * image otherwise; per LM iteration every chunk passes through the device twice
 * (Schur accumulation; back-substitution + trial-point linearisation), its
 * transfers overlapped with the kernels of the chunk before it.  Poses, the packed
 * partial S||rhs and the controller are resident per chunk, the dense image and x
 * once.  Same problem-construction calls, same options and iteration rows as
 * ba_solve; the trajectory equals the resident solve up to the summation order of
 * the chunks' partial sums.  ba_stream_finalize fails if a chunk does not fit the
 * arena ("use more chunks"). */
typedef struct ba_stream ba_stream;
int ba_stream_create(ba_stream **out, int device_id, int n_chunks, int64_t arena_bytes);
void ba_stream_destroy(ba_stream *s);
int ba_stream_set_cameras(ba_stream *s, int n_cam, const double *intr4, const double *T_cj12);
int ba_stream_set_poses(ba_stream *s, int n_pose, const double *T_jw12, const uint8_t *fixed);
int ba_stream_set_points(ba_stream *s, int n_pt, const double *X3, const uint8_t *fixed);
int ba_stream_set_observations(ba_stream *s, int64_t n_obs, const int32_t *cam, const int32_t *pose,
                               const int32_t *point, const double *uv2);
int ba_stream_finalize(ba_stream *s);
int ba_stream_solve(ba_stream *s, const ba_options *opt, ba_iter_info *out, int cap,
                    int *n_iter, int *converged);
/* the same loop in three pieces, like ba_lm_begin / ba_lm_iterate / ba_lm_sync */
int ba_stream_lm_begin(ba_stream *s, const ba_options *opt);
int ba_stream_lm_iterate(ba_stream *s, int n);
int ba_stream_lm_sync(ba_stream *s, ba_iter_info *out, int cap, int *n_iter, int *converged);
int ba_stream_get_poses(ba_stream *s, double *T_jw12);
int ba_stream_get_points(ba_stream *s, double *X3);
/* out6 = { device bytes of the arenas, bytes of the largest chunk, bytes of all
 * chunks together (what a resident solve would hold), bytes copied host->device and
 * device->host since ba_stream_finalize, number of chunks } */
int ba_stream_info(ba_stream *s, int64_t out6[6]);

/* ---- dense SPD solve alone (tests / micro-bench of the MFMA kernel) ---- */
/* Solves A x = b for symmetric positive (semi-)definite A (n x n row-major
 * host arrays) with the blocked fp64-MFMA Cholesky used for the reduced
 * camera system.  `ms` (optional) receives the device time of the solve. */
int ba_dense_spd_solve(ba_handle *h, int n, const double *A, const double *b,
                       double *x, double *ms);

/* ---- pose-only, monocular 6-DoF (fp32) --------------------------------- */
/* Solve_Monocular_6Dof, reference
 * core/pose_only_bundle_adjustment_solver.cpp:8-170.  T12 in/out is
 * reference_to_current_pose; mask is n bytes in/out (sticky false). */
typedef struct {
  float cost, cost_change, abs_step;
} ba_po_iter;
int ba_pose_only_mono6(ba_handle *h, const float *X3, const float *uv2, int n,
                       float fx, float fy, float cx, float cy, float *T12,
                       uint8_t *mask, const ba_options *opt, ba_po_iter *iters,
                       int cap, int *n_iter, int *converged,
                       float *debug_T12);

/* ---- pose-only, stereo 6-DoF (fp32) ------------------------------------- */
/* Solve_Stereo_6Dof, reference
 * core/pose_only_bundle_adjustment_solver.cpp:172-399.  intr_*4 = fx,fy,cx,cy;
 * T_lr12 = left_to_right_pose; T12 in/out = reference_to_current_left_pose;
 * a right pixel with a negative coordinate means "no right observation"
 * (:298); mask_l / mask_r are n bytes each, in/out (sticky false). */
int ba_pose_only_stereo6(ba_handle *h, const float *X3, const float *uvl2,
                         const float *uvr2, int n, const float *intr_l4,
                         const float *intr_r4, const float *T_lr12, float *T12,
                         uint8_t *mask_l, uint8_t *mask_r, const ba_options *opt,
                         ba_po_iter *iters, int cap, int *n_iter, int *converged,
                         float *debug_T12);

#ifdef __cplusplus
}
#endif
#endif /* BA_HIP_H_ */
