// Minimal stand-in for the part of the Ceres Solver API that the reference's
// pose-only comparison (reference test/test_compare_ceres_vs_native.cpp:178-205,
// core/pose_only_bundle_adjustment_solver_ceres.h:84-128) uses, written from
// scratch for this repository.  It exists ONLY because Ceres is not installed
// in this image or on the GPU box and cannot be fetched; with a real Ceres on
// the include path this directory must simply not be added (-I).
//
// What it is: CostFunction / SizedCostFunction / AutoDiffCostFunction (forward
// mode dual numbers, jet.h), LossFunction (+ TrivialLoss, HuberLoss,
// CauchyLoss), Problem (parameter blocks by pointer, constant blocks),
// Solver::Options / Solver::Summary and Solve() = a dense Levenberg-Marquardt
// trust-region minimiser of 1/2 sum rho(|r|^2) on the normal equations with the
// step-acceptance, radius-update and termination rules Ceres documents for its
// LEVENBERG_MARQUARDT strategy.  What it is not: Ceres.  No sparse algebra, no
// manifolds, no line search, no threading; problems whose total parameter
// dimension is small (the pose-only problems have 6).
// It serves as the INDEPENDENT fp64 check of the HIP pose-only solver: autodiff
// Jacobians, angle-axis parametrisation and trust-region LM share nothing with
// the analytic Gauss-Newton of reference
// core/pose_only_bundle_adjustment_solver.cpp:8-170.
#ifndef BA_SHIM_CERES_CERES_H_
#define BA_SHIM_CERES_CERES_H_

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "jet.h"

namespace ceres {

enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };

// ---------------------------------------------------------------------------
class CostFunction {
 public:
  virtual ~CostFunction() {}
  // jacobians[b] (may be null) is row-major num_residuals x block size b
  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
  const std::vector<int> &parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int> parameter_block_sizes_;
  int num_residuals_ = 0;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int>{Ns...};
  }
};

template <typename CostFunctor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, Ns...> {
  static constexpr int kBlocks = sizeof...(Ns);
  static constexpr int kParams = (0 + ... + Ns);
  using JetT = Jet<double, kParams>;

 public:
  explicit AutoDiffCostFunction(CostFunctor *functor, Ownership ownership = TAKE_OWNERSHIP)
      : functor_(functor), ownership_(ownership) {}
  ~AutoDiffCostFunction() override {
    if (ownership_ == DO_NOT_TAKE_OWNERSHIP) (void)functor_.release();
  }

  bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override {
    if (jacobians == nullptr) return Call(parameters, residuals, std::make_index_sequence<kBlocks>());
    const int sizes[kBlocks] = {Ns...};
    JetT x[kParams];
    const JetT *blocks[kBlocks];
    int off = 0;
    for (int b = 0; b < kBlocks; ++b) {
      blocks[b] = x + off;
      for (int k = 0; k < sizes[b]; ++k) x[off + k] = JetT(parameters[b][k], off + k);
      off += sizes[b];
    }
    JetT r[kNumResiduals];
    if (!Call(blocks, r, std::make_index_sequence<kBlocks>())) return false;
    for (int i = 0; i < kNumResiduals; ++i) residuals[i] = r[i].a;
    off = 0;
    for (int b = 0; b < kBlocks; ++b) {
      if (jacobians[b] != nullptr)
        for (int i = 0; i < kNumResiduals; ++i)
          for (int k = 0; k < sizes[b]; ++k) jacobians[b][i * sizes[b] + k] = r[i].v[off + k];
      off += sizes[b];
    }
    return true;
  }

 private:
  template <typename T, size_t... I>
  bool Call(T const *const *p, T *r, std::index_sequence<I...>) const {
    return (*functor_)(p[I]..., r);
  }
  std::unique_ptr<CostFunctor> functor_;
  Ownership ownership_;
};

// ---------------------------------------------------------------------------
// rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s) for s = |r|^2
class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double s, double rho[3]) const = 0;
};
class TrivialLoss : public LossFunction {
 public:
  void Evaluate(double s, double rho[3]) const override { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
};
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }
 private:
  double a_, b_;
};
class CauchyLoss : public LossFunction {
 public:
  explicit CauchyLoss(double a) : b_(a * a), c_(1.0 / (a * a)) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1.0 + s * c_, inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }
 private:
  double b_, c_;
};

// ---------------------------------------------------------------------------
class Problem {
 public:
  struct Options {
    Ownership cost_function_ownership = TAKE_OWNERSHIP;
    Ownership loss_function_ownership = TAKE_OWNERSHIP;
  };
  Problem() {}
  explicit Problem(const Options &o) : options_(o) {}
  Problem(const Problem &) = delete;
  Problem &operator=(const Problem &) = delete;
  ~Problem() {
    std::vector<const void *> freed;
    auto once = [&freed](const void *p) {
      if (p == nullptr || std::find(freed.begin(), freed.end(), p) != freed.end()) return false;
      freed.push_back(p);
      return true;
    };
    for (ResidualBlock &rb : residual_blocks_) {
      if (options_.cost_function_ownership == TAKE_OWNERSHIP && once(rb.cost)) delete rb.cost;
      if (options_.loss_function_ownership == TAKE_OWNERSHIP && once(rb.loss)) delete rb.loss;
    }
  }

  using ResidualBlockId = int;
  template <typename... Ts>
  ResidualBlockId AddResidualBlock(CostFunction *cost, LossFunction *loss, double *x0, Ts *...xs) {
    return AddResidualBlock(cost, loss, std::vector<double *>{x0, xs...});
  }
  ResidualBlockId AddResidualBlock(CostFunction *cost, LossFunction *loss, const std::vector<double *> &blocks) {
    ResidualBlock rb;
    rb.cost = cost;
    rb.loss = loss;
    const std::vector<int> &sizes = cost->parameter_block_sizes();
    for (size_t b = 0; b < blocks.size() && b < sizes.size(); ++b) {
      AddParameterBlock(blocks[b], sizes[b]);
      rb.blocks.push_back(blocks[b]);
    }
    residual_blocks_.push_back(rb);
    return static_cast<int>(residual_blocks_.size()) - 1;
  }
  void AddParameterBlock(double *values, int size) {
    auto it = parameter_blocks_.find(values);
    if (it == parameter_blocks_.end()) {
      ParameterBlock pb;
      pb.size = size;
      pb.order = static_cast<int>(parameter_blocks_.size());
      parameter_blocks_[values] = pb;
    }
  }
  void SetParameterBlockConstant(double *values) { parameter_blocks_.at(values).constant = true; }
  void SetParameterBlockVariable(double *values) { parameter_blocks_.at(values).constant = false; }
  int NumParameterBlocks() const { return static_cast<int>(parameter_blocks_.size()); }
  int NumResidualBlocks() const { return static_cast<int>(residual_blocks_.size()); }
  int NumResiduals() const {
    int n = 0;
    for (const ResidualBlock &rb : residual_blocks_) n += rb.cost->num_residuals();
    return n;
  }
  int NumParameters() const {
    int n = 0;
    for (const auto &kv : parameter_blocks_) n += kv.second.size;
    return n;
  }

 private:
  friend struct SolverImpl;
  struct ResidualBlock {
    CostFunction *cost = nullptr;
    LossFunction *loss = nullptr;
    std::vector<double *> blocks;
  };
  struct ParameterBlock {
    int size = 0, order = 0, offset = -1;
    bool constant = false;
  };
  Options options_;
  std::vector<ResidualBlock> residual_blocks_;
  std::map<double *, ParameterBlock> parameter_blocks_;
};

// ---------------------------------------------------------------------------
struct IterationSummary {
  int iteration = 0;
  bool step_is_successful = false;
  double cost = 0, cost_change = 0, gradient_max_norm = 0, step_norm = 0, relative_decrease = 0,
         trust_region_radius = 0;
};

class Solver {
 public:
  struct Options {
    int max_num_iterations = 50;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    double initial_trust_region_radius = 1e4;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    bool minimizer_progress_to_stdout = false;
    int num_threads = 1;
    LinearSolverType linear_solver_type = DENSE_NORMAL_CHOLESKY;
    TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
  };
  struct Summary {
    TerminationType termination_type = FAILURE;
    std::string message = "Solve was not called.";
    double initial_cost = -1, final_cost = -1;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    double total_time_in_seconds = 0;
    int num_parameters = 0, num_residuals = 0, num_residual_blocks = 0;
    std::vector<IterationSummary> iterations;
    bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE; }
    std::string BriefReport() const {
      char buf[512];
      std::snprintf(buf, sizeof(buf),
                    "Ceres-API stand-in Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, "
                    "Termination: %s",
                    num_successful_steps + num_unsuccessful_steps, initial_cost, final_cost,
                    termination_type == CONVERGENCE ? "CONVERGENCE"
                    : termination_type == NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE");
      return buf;
    }
    std::string FullReport() const {
      char buf[512];
      std::snprintf(buf, sizeof(buf),
                    "\nParameters %d, residual blocks %d, residuals %d\nSteps: %d successful, %d unsuccessful\n"
                    "Time: %.6f s\nMessage: %s\n",
                    num_parameters, num_residual_blocks, num_residuals, num_successful_steps, num_unsuccessful_steps,
                    total_time_in_seconds, message.c_str());
      return BriefReport() + buf;
    }
  };
};

// Dense Levenberg-Marquardt on the normal equations (see the header comment).
struct SolverImpl {
  // cost = 1/2 sum rho(|r|^2); optionally the Gauss-Newton matrix H = J^T J and
  // the gradient g = J^T r of the (robustified) residuals
  static bool Evaluate(const Problem &problem, const std::vector<double> &x, int n, double *cost,
                       std::vector<double> *H, std::vector<double> *g) {
    if (H) H->assign(static_cast<size_t>(n) * n, 0.0);
    if (g) g->assign(n, 0.0);
    double total = 0.0;
    std::vector<double> r, jac_store;
    std::vector<const double *> pp;
    std::vector<double *> jp;
    std::vector<int> offs;
    for (const Problem::ResidualBlock &rb : problem.residual_blocks_) {
      const int nr = rb.cost->num_residuals();
      const std::vector<int> &sizes = rb.cost->parameter_block_sizes();
      const size_t nb = rb.blocks.size();
      r.assign(nr, 0.0);
      pp.resize(nb);
      jp.assign(nb, nullptr);
      offs.resize(nb);
      size_t need = 0;
      for (size_t b = 0; b < nb; ++b) need += static_cast<size_t>(nr) * sizes[b];
      jac_store.assign(need, 0.0);
      size_t at = 0;
      for (size_t b = 0; b < nb; ++b) {
        const Problem::ParameterBlock &pb = problem.parameter_blocks_.at(rb.blocks[b]);
        offs[b] = pb.constant ? -1 : pb.offset;
        // current value: variable blocks live in x, constant ones in user memory
        pp[b] = pb.constant ? rb.blocks[b] : &x[pb.offset];
        if (H && !pb.constant) jp[b] = &jac_store[at];
        at += static_cast<size_t>(nr) * sizes[b];
      }
      if (!rb.cost->Evaluate(pp.data(), r.data(), H ? jp.data() : nullptr)) return false;
      double s = 0.0;
      for (int i = 0; i < nr; ++i) s += r[i] * r[i];
      double scale = 1.0;
      if (rb.loss) {
        double rho[3];
        rb.loss->Evaluate(s, rho);
        total += 0.5 * rho[0];
        scale = std::sqrt(rho[1]);  // first-order robustification of r and J
      } else {
        total += 0.5 * s;
      }
      if (!H) continue;
      for (size_t a = 0; a < nb; ++a) {
        if (offs[a] < 0) continue;
        const int sa = sizes[a];
        for (int i = 0; i < nr; ++i)
          for (int k = 0; k < sa; ++k) {
            const double ja = jp[a][i * sa + k] * scale;
            (*g)[offs[a] + k] += ja * (r[i] * scale);
            for (size_t b = 0; b < nb; ++b) {
              if (offs[b] < 0) continue;
              const int sb = sizes[b];
              double *row = &(*H)[static_cast<size_t>(offs[a] + k) * n + offs[b]];
              for (int l = 0; l < sb; ++l) row[l] += ja * (jp[b][i * sb + l] * scale);
            }
          }
      }
    }
    *cost = total;
    return true;
  }

  // A d = b for symmetric positive definite A (n x n row-major, destroyed)
  static bool CholeskySolve(std::vector<double> &A, std::vector<double> &b, int n) {
    for (int j = 0; j < n; ++j) {
      double d = A[static_cast<size_t>(j) * n + j];
      for (int k = 0; k < j; ++k) d -= A[static_cast<size_t>(j) * n + k] * A[static_cast<size_t>(j) * n + k];
      if (!(d > 0.0)) return false;
      d = std::sqrt(d);
      A[static_cast<size_t>(j) * n + j] = d;
      for (int i = j + 1; i < n; ++i) {
        double s = A[static_cast<size_t>(i) * n + j];
        for (int k = 0; k < j; ++k) s -= A[static_cast<size_t>(i) * n + k] * A[static_cast<size_t>(j) * n + k];
        A[static_cast<size_t>(i) * n + j] = s / d;
      }
    }
    for (int i = 0; i < n; ++i) {
      double s = b[i];
      for (int k = 0; k < i; ++k) s -= A[static_cast<size_t>(i) * n + k] * b[k];
      b[i] = s / A[static_cast<size_t>(i) * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = b[i];
      for (int k = i + 1; k < n; ++k) s -= A[static_cast<size_t>(k) * n + i] * b[k];
      b[i] = s / A[static_cast<size_t>(i) * n + i];
    }
    return true;
  }

  static void Run(const Solver::Options &opt, Problem *problem, Solver::Summary *sum) {
    const auto t0 = std::chrono::steady_clock::now();
    *sum = Solver::Summary();
    // variable blocks get offsets in registration order
    std::vector<std::pair<int, double *>> order;
    for (auto &kv : problem->parameter_blocks_) order.emplace_back(kv.second.order, kv.first);
    std::sort(order.begin(), order.end());
    int n = 0;
    for (auto &o : order) {
      Problem::ParameterBlock &pb = problem->parameter_blocks_[o.second];
      pb.offset = pb.constant ? -1 : n;
      if (!pb.constant) n += pb.size;
    }
    sum->num_parameters = problem->NumParameters();
    sum->num_residuals = problem->NumResiduals();
    sum->num_residual_blocks = problem->NumResidualBlocks();
    std::vector<double> x(n), x_try(n), H, g, A, d(n), delta(n);
    for (auto &o : order) {
      const Problem::ParameterBlock &pb = problem->parameter_blocks_[o.second];
      if (!pb.constant) std::copy(o.second, o.second + pb.size, x.begin() + pb.offset);
    }
    auto write_back = [&]() {
      for (auto &o : order) {
        const Problem::ParameterBlock &pb = problem->parameter_blocks_[o.second];
        if (!pb.constant) std::copy(x.begin() + pb.offset, x.begin() + pb.offset + pb.size, o.second);
      }
    };
    auto finish = [&](TerminationType t, const std::string &msg, double cost) {
      write_back();
      sum->termination_type = t;
      sum->message = msg;
      sum->final_cost = cost;
      sum->total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    double cost = 0.0;
    if (!Evaluate(*problem, x, n, &cost, &H, &g)) return finish(FAILURE, "initial residual evaluation failed", -1);
    sum->initial_cost = cost;
    if (n == 0) return finish(CONVERGENCE, "no variable parameter block", cost);
    double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
    auto gmax = [&]() {
      double m = 0.0;
      for (double v : g) m = std::max(m, std::fabs(v));
      return m;
    };
    if (opt.minimizer_progress_to_stdout)
      std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n"
                  "%4d % .6e % .2e % .2e % .2e % .2e % .2e\n", 0, cost, 0.0, gmax(), 0.0, 0.0, radius);
    if (gmax() <= opt.gradient_tolerance) return finish(CONVERGENCE, "gradient tolerance reached", cost);
    for (int it = 1; it <= opt.max_num_iterations; ++it) {
      // (J^T J + D^2 / radius) delta = -g,  D^2 = clamp(diag(J^T J))
      A = H;
      for (int k = 0; k < n; ++k) {
        const double dk = std::min(std::max(H[static_cast<size_t>(k) * n + k], opt.min_lm_diagonal * opt.min_lm_diagonal),
                                   opt.max_lm_diagonal * opt.max_lm_diagonal);
        A[static_cast<size_t>(k) * n + k] += dk / radius;
      }
      for (int k = 0; k < n; ++k) delta[k] = -g[k];
      IterationSummary is;
      is.iteration = it;
      is.trust_region_radius = radius;
      bool ok = CholeskySolve(A, delta, n);
      double model_change = 0.0, new_cost = cost;
      if (ok) {
        // model decrease = -(g . delta + 1/2 delta^T H delta)
        double gd = 0.0, dHd = 0.0;
        for (int a = 0; a < n; ++a) {
          gd += g[a] * delta[a];
          double row = 0.0;
          for (int b = 0; b < n; ++b) row += H[static_cast<size_t>(a) * n + b] * delta[b];
          dHd += delta[a] * row;
        }
        model_change = -(gd + 0.5 * dHd);
        ok = model_change > 0.0;
      }
      double step_norm = 0.0, x_norm = 0.0;
      if (ok) {
        for (int k = 0; k < n; ++k) {
          x_try[k] = x[k] + delta[k];
          step_norm += delta[k] * delta[k];
          x_norm += x[k] * x[k];
        }
        step_norm = std::sqrt(step_norm);
        x_norm = std::sqrt(x_norm);
        ok = Evaluate(*problem, x_try, n, &new_cost, nullptr, nullptr) && std::isfinite(new_cost);
      }
      const double rel = ok ? (cost - new_cost) / model_change : -1.0;
      is.step_norm = step_norm;
      is.relative_decrease = rel;
      if (ok && step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
        sum->iterations.push_back(is);
        return finish(CONVERGENCE, "parameter tolerance reached", cost);
      }
      if (ok && rel > opt.min_relative_decrease) {
        const double change = cost - new_cost;
        x = x_try;
        if (!Evaluate(*problem, x, n, &new_cost, &H, &g)) return finish(FAILURE, "residual evaluation failed", cost);
        cost = new_cost;
        is.step_is_successful = true;
        is.cost = cost;
        is.cost_change = change;
        is.gradient_max_norm = gmax();
        ++sum->num_successful_steps;
        sum->iterations.push_back(is);
        radius = std::min(opt.max_trust_region_radius,
                          radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
        decrease_factor = 2.0;
        if (opt.minimizer_progress_to_stdout)
          std::printf("%4d % .6e % .2e % .2e % .2e % .2e % .2e\n", it, cost, change, is.gradient_max_norm, step_norm,
                      rel, radius);
        if (is.gradient_max_norm <= opt.gradient_tolerance)
          return finish(CONVERGENCE, "gradient tolerance reached", cost);
        if (std::fabs(change) <= opt.function_tolerance * cost)
          return finish(CONVERGENCE, "function tolerance reached", cost);
      } else {
        ++sum->num_unsuccessful_steps;
        is.cost = cost;
        sum->iterations.push_back(is);
        radius /= decrease_factor;
        decrease_factor *= 2.0;
        if (opt.minimizer_progress_to_stdout)
          std::printf("%4d % .6e % .2e % .2e % .2e % .2e % .2e  (rejected)\n", it, cost, 0.0, gmax(), step_norm, rel,
                      radius);
        if (radius <= opt.min_trust_region_radius)
          return finish(CONVERGENCE, "trust region radius below its minimum", cost);
      }
    }
    finish(NO_CONVERGENCE, "maximum number of iterations reached", cost);
  }
};

inline void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary) {
  SolverImpl::Run(options, problem, summary);
}

}  // namespace ceres
#endif
