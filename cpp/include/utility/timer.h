// Wall-clock helpers exposing the interface the reference's callers use
// (utility/timer.h:8-34): free tic()/toc() and a named StopWatch with laps.
#ifndef BA_FACADE_TIMER_H_
#define BA_FACADE_TIMER_H_

#include <chrono>
#include <string>

namespace timer {

void tic();                            // start the process-wide timer
double toc(bool print_elapsed);        // milliseconds since tic()
const std::string currentDateTime();   // "yyyy-mm-dd.hh:mm:ss"

class StopWatch {
  using SteadyClock = std::chrono::high_resolution_clock;

 public:
  explicit StopWatch(const std::string &name);
  ~StopWatch();
  // all four return milliseconds; the flag prints the value with the watch's name
  double Start(const bool print_elapsed = false);                 // (re)starts, returns 0
  double GetLapTimeFromStart(const bool print_elapsed = false);   // since Start()
  double GetLapTimeFromLatest(const bool print_elapsed = false);  // since the previous lap / Start()
  double Stop(const bool print_elapsed = false);                  // since Start(), freezes the watch

 private:
  std::string timer_name_;
  SteadyClock::time_point start_, intermediate_, end_;
};

}  // namespace timer
#endif
