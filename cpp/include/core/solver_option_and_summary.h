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

enum class SolverType { UNDEFINED = -1, GRADIENT_DESCENT = 0, GAUSS_NEWTON = 1, LEVENBERG_MARQUARDT = 2 };
enum class IterationStatus { UNDEFINED = -1, UPDATE = 0, UPDATE_TRUST_MORE = 1, SKIPPED = 2 };

struct OptimizationInfo {
  double cost{-1.0};
  double cost_change{-1.0};
  double average_reprojection_error{-1.0};
  double abs_gradient{-1.0};
  double abs_step{-1.0};
  double damping_term{-1.0};
  double iter_time{-1.0};
  IterationStatus iteration_status{IterationStatus::UNDEFINED};
};

class Options {
  friend class PoseOnlyBundleAdjustmentSolver;
  friend class FullBundleAdjustmentSolver;

 public:
  Options() {}
  ~Options() {}

  SolverType solver_type{SolverType::GAUSS_NEWTON};  // ignored by full BA
  struct {
    float threshold_step_size{1e-5};
    float threshold_cost_change{1e-5};
  } convergence_handle;
  struct {
    float threshold_huber_loss{1.0};
    float threshold_outlier_rejection{2.0};
  } outlier_handle;
  struct {
    int max_num_iterations{50};
  } iteration_handle;
  struct {
    float initial_lambda{100.0};
    float decrease_ratio_lambda{0.33f};
    float increase_ratio_lambda{3.0f};
  } trust_region_handle;
};

class Summary {
  friend class PoseOnlyBundleAdjustmentSolver;
  friend class FullBundleAdjustmentSolver;

 public:
  Summary();
  ~Summary();
  std::string BriefReport();
  std::string FullReport();  // declared but never defined by the reference
  const double GetTotalTimeInSecond() const;
  // read access for tests (not in the reference)
  const std::vector<OptimizationInfo> &GetOptimizationInfoList() const { return optimization_info_list_; }
  bool IsConverged() const { return convergence_status_; }

 protected:
  std::vector<OptimizationInfo> optimization_info_list_;
  int max_iteration_{0};
  double total_time_in_millisecond_{0.0};
  double threshold_step_size_{0.0};
  double threshold_cost_change_{0.0};
  bool convergence_status_{false};
};

}  // namespace analytic_solver
}  // namespace visual_navigation
#endif
