#include "utility/timer.h"

#include <ctime>
#include <iostream>

namespace timer {
namespace {
std::chrono::high_resolution_clock::time_point g_tic;
double ms_between(std::chrono::high_resolution_clock::time_point a, std::chrono::high_resolution_clock::time_point b) {
  return std::chrono::duration<double, std::milli>(b - a).count();
}
}  // namespace

void tic() { g_tic = std::chrono::high_resolution_clock::now(); }
double toc(bool flag_verbose) {
  const double ms = ms_between(g_tic, std::chrono::high_resolution_clock::now());
  if (flag_verbose) std::cout << "elapsed: " << ms << " [ms]\n";
  return ms;
}
const std::string currentDateTime() {
  const std::time_t now = std::time(nullptr);
  char buf[32];
  std::strftime(buf, sizeof(buf), "%Y-%m-%d.%H:%M:%S", std::localtime(&now));
  return buf;
}

StopWatch::StopWatch(const std::string &stopwatch_name) : timer_name_(stopwatch_name) {}
StopWatch::~StopWatch() {}
double StopWatch::Start(const bool) {
  start_ = intermediate_ = SteadyClock::now();
  return 0.0;
}
double StopWatch::GetLapTimeFromStart(const bool flag_verbose) {
  intermediate_ = SteadyClock::now();
  const double ms = ms_between(start_, intermediate_);
  if (flag_verbose) std::cout << timer_name_ << ": " << ms << " [ms] from start\n";
  return ms;
}
double StopWatch::GetLapTimeFromLatest(const bool flag_verbose) {
  const SteadyClock::time_point now = SteadyClock::now();
  const double ms = ms_between(intermediate_, now);
  intermediate_ = now;
  if (flag_verbose) std::cout << timer_name_ << ": " << ms << " [ms] lap\n";
  return ms;
}
double StopWatch::Stop(const bool flag_verbose) {
  end_ = SteadyClock::now();
  const double ms = ms_between(start_, end_);
  if (flag_verbose) std::cout << timer_name_ << ": " << ms << " [ms] total\n";
  return ms;
}
}  // namespace timer
