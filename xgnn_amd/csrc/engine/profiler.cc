// profiler.cc -- step / epoch / init log store and Chrome-trace dump.
// Reference: samgraph/common/profiler.{h,cc} (LogStep/LogStepAdd/LogEpochAdd/LogInit :166-215,
// trace JSON :349-380).  Only the storage and the read-back the example scripts use is kept;
// item codes are the reference's enum values (profiler.h:30-163).
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "engine.h"

namespace sam {

void Profiler::Resize(size_t num_epoch, size_t num_step) {
  std::lock_guard<std::mutex> lk(mu_);
  num_epoch_ = num_epoch ? num_epoch : 1;
  num_step_ = num_step ? num_step : 1;
  step_.assign(num_epoch_ * num_step_ * kMaxStep, 0.0);
  epoch_.assign(num_epoch_ * kMaxEpoch, 0.0);
}

void Profiler::LogStep(uint64_t key, int item, double v) {
  if (item < 0 || item >= kMaxStep || key >= num_epoch_ * num_step_) return;
  step_[key * kMaxStep + item] = v;
}
void Profiler::LogStepAdd(uint64_t key, int item, double v) {
  if (item < 0 || item >= kMaxStep || key >= num_epoch_ * num_step_) return;
  step_[key * kMaxStep + item] += v;
}
void Profiler::LogEpochAdd(uint64_t key, int item, double v) {
  const uint64_t epoch = key / num_step_; // GetEpochFromKey, engine.h:53
  if (item < 0 || item >= kMaxEpoch || epoch >= num_epoch_) return;
  std::lock_guard<std::mutex> lk(mu_);
  epoch_[epoch * kMaxEpoch + item] += v;
}
double Profiler::GetStep(uint64_t key, int item) const {
  if (item < 0 || item >= kMaxStep || key >= num_epoch_ * num_step_) return 0.0;
  return step_[key * kMaxStep + item];
}
double Profiler::GetEpoch(uint64_t epoch, int item) const {
  if (item < 0 || item >= kMaxEpoch || epoch >= num_epoch_) return 0.0;
  return epoch_[epoch * kMaxEpoch + item];
}

void Profiler::Trace(uint64_t key, int item, uint64_t ts, bool begin) {
  std::lock_guard<std::mutex> lk(mu_);
  if (begin) {
    trace_.push_back({key, item, ts, 0});
  } else {
    for (auto it = trace_.rbegin(); it != trace_.rend(); ++it)
      if (it->key == key && it->item == item && it->end == 0) { it->end = ts; break; }
  }
}

void Profiler::DumpTrace() { // Chrome trace events to stderr, as samgraph_dump_trace does (operation.cc:502-505,
  // profiler.cc:349-380); only when SAMGRAPH_DUMP_TRACE is set (run_config.cc:130-132)
  if (!getenv("SAMGRAPH_DUMP_TRACE")) return;
  std::lock_guard<std::mutex> lk(mu_);
  std::fprintf(stderr, "[");
  bool first = true;
  for (auto &r : trace_) {
    if (!r.end) continue;
    std::fprintf(stderr, "%s{\"name\":\"item%d\",\"ph\":\"X\",\"pid\":0,\"tid\":%d,\"ts\":%llu,\"dur\":%llu,\"args\":{\"key\":%llu}}",
                 first ? "" : ",", r.item, r.item, (unsigned long long)r.begin, (unsigned long long)(r.end - r.begin),
                 (unsigned long long)r.key);
    first = false;
  }
  std::fprintf(stderr, "]\n");
}

void Profiler::ReportStep(uint64_t epoch, uint64_t step) {
  const uint64_t key = epoch * num_step_ + step;
  std::printf("    [Step(%llu, %llu)] L1 sample %.4f | copy %.4f | #node %.0f | #sample %.0f | feat MB %.2f | miss MB %.2f\n",
              (unsigned long long)epoch, (unsigned long long)step, GetStep(key, 3), GetStep(key, 6), GetStep(key, 1),
              GetStep(key, 0), GetStep(key, 9) / 1e6, GetStep(key, 13) / 1e6);
}

void Profiler::ReportEpoch(uint64_t epoch) {
  std::printf("  [Epoch %llu] sample %.4f s | copy %.4f s | #sample %.0f | feat GB %.3f | miss GB %.3f\n",
              (unsigned long long)epoch, GetEpoch(epoch, 0), GetEpoch(epoch, 8), GetEpoch(epoch, 15),
              GetEpoch(epoch, 12) / 1e9, GetEpoch(epoch, 13) / 1e9);
}

} // namespace sam
