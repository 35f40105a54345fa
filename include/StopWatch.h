// StopWatch.h — accumulating stop-watch with the reference's four-call interface
// (include/StopWatch.h:13-33: Reset / Resume / Pause / GetTime, seconds as float).
//
// Deliberate difference: the reference reads clock() = process CPU time (include/StopWatch.h:45,52),
// which under-reports a host thread that sleeps on a GPU sync.  This one reads
// the monotonic wall clock, the only meaningful clock around device work; per-layer device times come
// from HIP events (CaffeEva::DispElpsTime).
#ifndef QCNN_HOST_STOPWATCH_H_
#define QCNN_HOST_STOPWATCH_H_

#include <chrono>

class StopWatch {
 public:
  StopWatch() : running_(false), total_(0.0f) {}
  inline void Reset(void) { running_ = false; total_ = 0.0f; }
  inline void Resume(void) {
    if (running_) return;
    running_ = true;
    begin_ = std::chrono::steady_clock::now();
  }
  inline void Pause(void) {
    if (!running_) return;
    running_ = false;
    total_ += std::chrono::duration<float>(std::chrono::steady_clock::now() - begin_).count();
  }
  inline float GetTime(void) { return total_; }
  // extension: add externally measured seconds (device event times)
  inline void AddSeconds(float s) { total_ += s; }

 private:
  bool running_;
  float total_;
  std::chrono::steady_clock::time_point begin_;
};

#endif  // QCNN_HOST_STOPWATCH_H_
