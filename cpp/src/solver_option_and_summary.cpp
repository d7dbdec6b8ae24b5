// Summary reporting (reference core/solver_option_and_summary.cpp:8-84).
#include "core/solver_option_and_summary.h"

#include <cstdio>

namespace visual_navigation {
namespace analytic_solver {

Summary::Summary() {}
Summary::~Summary() {}

const double Summary::GetTotalTimeInSecond() const { return total_time_in_millisecond_ * 0.001; }

std::string Summary::FullReport() { return BriefReport(); }

std::string Summary::BriefReport() {
  std::string out =
      "itr   total_cost   avg.reproj.  cost_change  |step|   |gradient|  damp_term  itr_time[ms] itr_stat\n";
  char line[256];
  int it = 0;
  for (const OptimizationInfo &info : optimization_info_list_) {
    std::snprintf(line, sizeof(line), "%3d  %.6e    %.2e    %.2e   %.2e   %.2e    %.2e   %.2e     ", it++, info.cost,
                  info.average_reprojection_error, info.cost_change, info.abs_step, info.abs_gradient,
                  info.damping_term, info.iter_time);
    out += line;
    switch (info.iteration_status) {
      case IterationStatus::UPDATE: out += "UPDATE"; break;
      case IterationStatus::SKIPPED: out += TEXT_YELLOW(" SKIP "); break;
      case IterationStatus::UPDATE_TRUST_MORE: out += TEXT_GREEN("UPDATE"); break;
      default: break;
    }
    out += "\n";
  }
  const size_t n = optimization_info_list_.size();
  std::ostringstream ss;
  ss << std::setprecision(5);
  ss << "Analytic Solver Report:\n";
  ss << "  Iterations      : " << n << "\n";
  ss << "  Total time      : " << total_time_in_millisecond_ * 0.001 << " [second]\n";
  if (n > 0) {  // the reference dereferences front()/back() even when empty
    ss << "  Initial cost    : " << optimization_info_list_.front().cost << "\n";
    ss << "  Final cost      : " << optimization_info_list_.back().cost << "\n";
    ss << "  Initial reproj. : " << optimization_info_list_.front().average_reprojection_error << " [pixel]\n";
    ss << "  Final reproj.   : " << optimization_info_list_.back().average_reprojection_error << " [pixel]\n";
  }
  ss << ", Termination     : " << (convergence_status_ ? TEXT_GREEN("CONVERGENCE") : TEXT_YELLOW("NO_CONVERGENCE"))
     << "\n";
  if (max_iteration_ == static_cast<int>(n))
    ss << TEXT_YELLOW(" WARNIING: MAX ITERATION is reached ! The solution could be local minima.\n");
  return out + ss.str();
}

}  // namespace analytic_solver
}  // namespace visual_navigation
