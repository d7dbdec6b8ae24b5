// Options / Summary surface of the analytic solvers, re-authored for the HIP
// path.  Field and type names follow the reference
// (core/solver_option_and_summary.h:25-93) so that code written against it
// compiles unchanged; the reference's stray `#include "ceres/ceres.h"` is
// dropped (nothing here needs Ceres).
#ifndef BA_FACADE_SOLVER_OPTION_AND_SUMMARY_H_
#define BA_FACADE_SOLVER_OPTION_AND_SUMMARY_H_

#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace visual_navigation {
namespace analytic_solver {

// ANSI colour helpers (same macro names as the reference, :13-23)
#define BA_FACADE_COLOUR(code, str) \
  (std::string("\033[0;" code "m") + str + std::string("\033[0m"))
#define TEXT_RED(str) BA_FACADE_COLOUR("31", str)
#define TEXT_GREEN(str) BA_FACADE_COLOUR("32", str)
#define TEXT_YELLOW(str) BA_FACADE_COLOUR("33", str)
#define TEXT_BLUE(str) BA_FACADE_COLOUR("34", str)
#define TEXT_MAGENTA(str) BA_FACADE_COLOUR("35", str)
#define TEXT_CYAN(str) BA_FACADE_COLOUR("36", str)

// Which update rule a solver runs.  FullBundleAdjustmentSolver::Solve is always
// Levenberg-Marquardt whatever this says; FullBundleAdjustmentSolverRefactor obeys it.
enum class SolverType : int { UNDEFINED = -1, GRADIENT_DESCENT = 0, GAUSS_NEWTON = 1, LEVENBERG_MARQUARDT = 2 };
// Outcome of one iteration as logged in the Summary.
enum class IterationStatus : int { UNDEFINED = -1, UPDATE = 0, UPDATE_TRUST_MORE = 1, SKIPPED = 2 };

// One row of the iteration log (filled from ba_iter_info / ba_po_iter).
struct OptimizationInfo {
  double cost = -1.0;                        // sum of residual norms after the step (0.01-pixel units)
  double cost_change = -1.0;                 // |cost - previous cost|
  double average_reprojection_error = -1.0;  // cost / #observations
  double abs_gradient = -1.0;                // unused by the solvers (always 0)
  double abs_step = -1.0;                    // mean parameter step norm
  double damping_term = -1.0;                // lambda after the step (-1: pose-only)
  double iter_time = -1.0;                   // [ms]
  IterationStatus iteration_status = IterationStatus::UNDEFINED;
};

// Tunables.  All real-valued fields are float, as callers of the reference
// assign float literals to them; they are promoted inside the solvers.
class Options {
 public:
  struct ConvergenceHandle {
    float threshold_step_size = 1e-5f;    // stop when the mean step norm falls below
    float threshold_cost_change = 1e-5f;  // ... or the cost changes by less than this
  };
  struct OutlierHandle {
    float threshold_huber_loss = 1.0f;         // |r_u| + |r_v| above which the weight decays
    float threshold_outlier_rejection = 2.0f;  // pose-only: inlier mask threshold [pixel]
  };
  struct IterationHandle {
    int max_num_iterations = 50;
  };
  struct TrustRegionHandle {
    float initial_lambda = 100.0f;
    float decrease_ratio_lambda = 0.33f;  // applied when the gain ratio exceeds 0.5
    float increase_ratio_lambda = 3.0f;   // applied when the step is rejected
  };

  SolverType solver_type = SolverType::GAUSS_NEWTON;
  ConvergenceHandle convergence_handle;
  OutlierHandle outlier_handle;
  IterationHandle iteration_handle;
  TrustRegionHandle trust_region_handle;
};

// Result of a Solve call: the iteration log plus totals; printed by BriefReport().
class Summary {
 public:
  Summary();
  ~Summary();
  std::string BriefReport();                  // table of the iteration log + termination line
  std::string FullReport();                   // (the reference declares it and never defines it)
  const double GetTotalTimeInSecond() const;  // wall time of Solve
  // read access for tests (not in the reference)
  const std::vector<OptimizationInfo> &GetOptimizationInfoList() const { return optimization_info_list_; }
  bool IsConverged() const { return convergence_status_; }

 protected:
  friend class FullBundleAdjustmentSolver;
  friend class PoseOnlyBundleAdjustmentSolver;
  std::vector<OptimizationInfo> optimization_info_list_;  // one entry per logged iteration
  double total_time_in_millisecond_ = 0.0;
  double threshold_step_size_ = 0.0, threshold_cost_change_ = 0.0;  // echoed in the report
  int max_iteration_ = 0;
  bool convergence_status_ = false;
};

}  // namespace analytic_solver
}  // namespace visual_navigation
#endif
