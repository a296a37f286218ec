// team.h -- a persistent thread team with OpenMP's static `parallel for` split (host-side loops of the engine).
//
// The reference's host loops are `#pragma omp parallel for num_threads(RunConfig::omp_thread_num)` with the default
// static schedule: thread t takes one contiguous block, thread 0 is the caller, and per-thread state (the
// `static thread_local` generator of RandomID) lives as long as the thread.  This team reproduces exactly that without
// depending on an OpenMP runtime: used by the arch0 deployment (cpu_engine.cc) and by the host-staged miss extract
// of arch6 (engine.cc, `gpu_extract` off).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sam {

class Team {
 public:
  explicit Team(int n) : n_(std::max(1, n)) {
    for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t] { Loop(t); });
  }
  ~Team() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
  }
  int size() const { return n_; }
  // body(lo, hi, tid): thread tid takes iterations [lo, hi) -- the first n % T threads one iteration more
  void ParallelFor(size_t n, const std::function<void(size_t, size_t, int)> &body) {
    if (n_ == 1) {
      body(0, n, 0);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &body;
      job_n_ = n;
      pending_ = n_ - 1;
      ++gen_;
    }
    cv_.notify_all();
    Run(0, n, body);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void Run(int tid, size_t n, const std::function<void(size_t, size_t, int)> &body) const {
    const size_t q = n / n_, r = n % n_;
    const size_t lo = tid * q + std::min<size_t>(tid, r), hi = lo + q + ((size_t)tid < r ? 1 : 0);
    if (lo < hi) body(lo, hi, tid);
  }
  void Loop(int tid) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(size_t, size_t, int)> *job;
      size_t n;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        job = job_;
        n = job_n_;
      }
      Run(tid, n, *job);
      {
        std::lock_guard<std::mutex> lk(mu_);
        --pending_;
      }
      done_cv_.notify_one();
    }
  }
  const int n_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(size_t, size_t, int)> *job_ = nullptr;
  size_t job_n_ = 0;
  int pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

} // namespace sam
