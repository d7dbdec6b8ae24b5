// Wall-clock helpers with the reference's interface (utility/timer.h:8-34).
#ifndef BA_FACADE_TIMER_H_
#define BA_FACADE_TIMER_H_

#include <chrono>
#include <string>

namespace timer {
void tic();
double toc(bool flag_verbose);        // ms since tic()
const std::string currentDateTime();  // yyyy-mm-dd.hh:mm:ss

class StopWatch {
 public:
  explicit StopWatch(const std::string &stopwatch_name);
  ~StopWatch();
  double Start(const bool flag_verbose = false);
  double GetLapTimeFromStart(const bool flag_verbose = false);
  double GetLapTimeFromLatest(const bool flag_verbose = false);
  double Stop(const bool flag_verbose = false);

 private:
  typedef std::chrono::high_resolution_clock Clock;
  std::string timer_name_;
  Clock::time_point start_, intermediate_, end_;
};
}  // namespace timer
#endif
