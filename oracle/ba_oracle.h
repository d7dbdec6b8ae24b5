/*
 * ba_oracle.h — C ABI of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a dependency-free, single-threaded CPU restatement of the
 * reference's FullBundleAdjustmentSolver::Solve hot path
 * (reference: core/full_bundle_adjustment_solver.cpp:381-1082) and of
 * PoseOnlyBundleAdjustmentSolver::Solve_Monocular_6Dof
 * (reference: core/pose_only_bundle_adjustment_solver.cpp:8-170).
 *
 * PARITY UNPINNED: the reference holds no golden vectors / assertions
 *  and cannot be built in this image (Eigen/Ceres/OpenCV
 * absent), so this restatement is pinned only by independent properties
 * (finite-difference Jacobians, convergence to ground truth on noise-free
 * scenes, LDLT residual checks) — see tests/test_oracle_*.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library. All values are in the solver's SCALED units (the facade
 * applies the reference's 0.01 scaling and pose inversion before calling).
 */
#ifndef BA_ORACLE_H_
#define BA_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ba_oracle ba_oracle;

/* Options mirror reference core/solver_option_and_summary.h:50-71 (fields are
 * float there; the float->double promotion is part of the arithmetic). */
typedef struct {
  float threshold_step_size;
  float threshold_cost_change;
  float threshold_huber_loss;
  float threshold_outlier_rejection;
  int max_num_iterations;
  float initial_lambda;
  float decrease_ratio_lambda;
  float increase_ratio_lambda;
  /* 1: plain Gauss-Newton of the refactored solver (reference
   * core/full_bundle_adjustment_solver_refactor.cpp:976-982): every step is
   * accepted, lambda stays at initial_lambda.  0: Levenberg-Marquardt. */
  int gauss_newton;
} ba_oracle_options;

/* One row per LM iteration; mirrors OptimizationInfo
 * (reference core/solver_option_and_summary.h:37-46) plus rho/model. */
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
  double trial_cost; /* cost at trial parameters (before SKIPPED overwrite) */
} ba_oracle_iter;

/* Problem description (copied).  cam_T / pose_T are 12 doubles: row-major
 * 3x3 rotation followed by translation.  Observation order is the insertion
 * order (it decides the surviving B_ji, reference :826). */
ba_oracle *ba_oracle_create(int n_cam, const double *cam_intr4,
                            const double *cam_T12, int n_pose,
                            const double *pose_T12, const uint8_t *pose_fixed,
                            int n_pt, const double *pt_X3,
                            const uint8_t *pt_fixed, int64_t n_obs,
                            const int32_t *obs_cam, const int32_t *obs_pose,
                            const int32_t *obs_pt, const double *obs_uv2);
void ba_oracle_destroy(ba_oracle *o);

/* dense_faithful != 0: additionally allocate and re-zero the reference's
 * dense N x M block grids every iteration (timing fidelity only). */
void ba_oracle_set_dense_faithful(ba_oracle *o, int on);

/* fast_solve != 0: the reduced system is solved by an envelope LDL^T without
 * pivoting instead of the restated Eigen pivoted LDLT (same x up to roundoff
 * for positive definite S; cross-checked in tests/test_oracle_pins.py).  A
 * TEST shortcut for long trajectories at the BASELINE sizes, where the
 * reference-style dense factorisation takes 17 s per iteration; not a
 * restatement of reference arithmetic. */
void ba_oracle_set_fast_solve(ba_oracle *o, int on);

int ba_oracle_num_opt_poses(const ba_oracle *o);
int ba_oracle_num_opt_points(const ba_oracle *o);
int64_t ba_oracle_num_pairs(const ba_oracle *o);

/* --- stage entry points (reference line ranges in ba_oracle.cpp) --- */
double ba_oracle_cost(ba_oracle *o);                        /* :381-433 */
void ba_oracle_linearize(ba_oracle *o, double huber);       /* :716-831 */
void ba_oracle_damp_invert(ba_oracle *o, double lambda);    /* :833-856 */
void ba_oracle_schur(ba_oracle *o);                         /* :858-888 */
void ba_oracle_solve_reduced(ba_oracle *o);                 /* :890-908 */
void ba_oracle_backsub(ba_oracle *o);                       /* :910-917 */
void ba_oracle_backup(ba_oracle *o);                        /* :457-469 */
void ba_oracle_revert(ba_oracle *o);                        /* :470-482 */
void ba_oracle_update(ba_oracle *o);                        /* :484-500 */
double ba_oracle_model_change(ba_oracle *o);                /* :435-455 */
void ba_oracle_step_norms(ba_oracle *o, double *pose_norm_sum,
                          double *point_norm_sum);          /* :960-963 */

/* Full LM loop (:705-1008). Returns number of iterations run. */
int ba_oracle_solve(ba_oracle *o, const ba_oracle_options *opt,
                    ba_oracle_iter *iters, int cap, int *converged);

/* Per-stage wall time of the last ba_oracle_solve call, ms, summed over
 * iterations: [0] build, [1] schur(+damp/invert), [2] solve+backsub,
 * [3] control (update, cost, model). */
void ba_oracle_stage_ms(const ba_oracle *o, double out4[4]);

/* --- readers (opt-index order = input order of non-fixed entries) --- */
void ba_oracle_get_poses(const ba_oracle *o, double *pose_T12 /* n_pose*12 */);
void ba_oracle_get_points(const ba_oracle *o, double *pt_X3 /* n_pt*3 */);
void ba_oracle_get_A(const ba_oracle *o, double *A36, double *a6);
void ba_oracle_get_C(const ba_oracle *o, double *C9, double *b3);
void ba_oracle_get_Cinv(const ba_oracle *o, double *Cinv9, double *Cinvb3);
/* pairs sorted by (i_opt, j_opt); W = B_ji, 6x3 row-major */
void ba_oracle_get_pairs(const ba_oracle *o, int32_t *pair_i, int32_t *pair_j,
                         double *W18);
void ba_oracle_get_S(const ba_oracle *o, double *S /* (6N)^2 row-major */,
                     double *rhs /* 6N */);
void ba_oracle_get_xy(const ba_oracle *o, double *x6, double *y3);
/* Overwrite S||rhs / x (tests of the multi-GPU exchange protocol: the reduced
 * system summed over shards is injected before the dense solve). */
void ba_oracle_set_S(ba_oracle *o, const double *S, const double *rhs);
void ba_oracle_set_x(ba_oracle *o, const double *x6);

/* Eigen-style pivoted LDLT solve of a dense symmetric system (lower part
 * read), nrhs right-hand sides, column-major rhs. Exposed for tests. */
void ba_oracle_ldlt_solve(int n, const double *A_rowmajor, int nrhs,
                          const double *B_colmajor, double *X_colmajor);

/* --- pose-only, monocular 6-DoF (fp32), reference
 * core/pose_only_bundle_adjustment_solver.cpp:8-170 --- */
typedef struct {
  float cost, cost_change, abs_step;
} ba_oracle_po_iter;
/* T12 (in/out): reference_to_current pose, row-major R then t (fp32).
 * mask: n bytes (in/out, sticky false). Returns 1 on success (0 = NaN). */
int ba_oracle_pose_only_mono6(const float *X3, const float *uv2, int n,
                              float fx, float fy, float cx, float cy,
                              float *T12, uint8_t *mask,
                              const ba_oracle_options *opt,
                              ba_oracle_po_iter *iters, int cap, int *n_iter,
                              int *converged, float *debug_T12 /* cap*12 or NULL */);

/* --- pose-only, stereo 6-DoF (fp32), reference
 * core/pose_only_bundle_adjustment_solver.cpp:172-399 --- */
/* intr_*4 = fx,fy,cx,cy; T_lr12 = left_to_right_pose; a right pixel with a
 * negative coordinate marks "no right observation" (:298). */
int ba_oracle_pose_only_stereo6(const float *X3, const float *uvl2,
                                const float *uvr2, int n, const float *intr_l4,
                                const float *intr_r4, const float *T_lr12,
                                float *T12, uint8_t *mask_l, uint8_t *mask_r,
                                const ba_oracle_options *opt,
                                ba_oracle_po_iter *iters, int cap, int *n_iter,
                                int *converged, float *debug_T12);

#ifdef __cplusplus
}
#endif
#endif
