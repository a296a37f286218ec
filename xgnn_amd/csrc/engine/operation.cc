// operation.cc -- extern "C" samgraph_* entry points (include/samgraph.h).
// Reference: samgraph/common/operation.cc:49-584 and samgraph/torch/adapter.cc:62-193.
#include <sys/wait.h>

#include <chrono>
#include <cstring>

#include "engine.h"

using sam::Engine;

static uint64_t now_us() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
             std::chrono::system_clock::now().time_since_epoch()).count();
}

static void fill(samgraph_tensor_t *t, void *data, int64_t d0, int64_t d1, int ndim, int dtype, int dev_type, int dev_id) {
  t->data = data; t->shape[0] = d0; t->shape[1] = d1; t->ndim = ndim; t->dtype = dtype;
  t->device_type = dev_type; t->device_id = dev_id;
}

extern "C" {

void samgraph_config(const char **keys, const char **values, const size_t n) {
  std::unordered_map<std::string, std::string> kv;
  for (size_t i = 0; i < n; ++i) kv[keys[i]] = values[i];
  Engine::Get().Configure(kv);
}
void samgraph_init(void) { Engine::Get().Init(); }
void samgraph_start(void) { Engine::Get().Start(); }
void samgraph_shutdown(void) { Engine::Get().Shutdown(); }
void samgraph_data_init(void) { Engine::Get().DataInit(); }
void samgraph_sample_init(int worker_id, const char *ctx) { Engine::Get().SampleInit(worker_id, ctx); }
void samgraph_train_init(int worker_id, const char *ctx) { Engine::Get().TrainInit(worker_id, ctx); }
void samgraph_extract_start(int count) { Engine::Get().ExtractStart(count); }
void samgraph_forward_barrier(void) { Engine::Get().Barrier(); }
void samgraph_um_sample_init(int) { sam::fatal(__FILE__, __LINE__, "arch9 (unified-memory sampling) is not part of this build"); }
void samgraph_switch_init(int, const char *, double) { sam::fatal(__FILE__, __LINE__, "arch5 (switcher) is not part of this build"); }

int samgraph_wait_one_child(void) { // operation.cc:573-584
  int st = 0;
  pid_t pid = waitpid(-1, &st, 0);
  (void)pid;
  if (WIFSIGNALED(st)) return 1;
  if (WEXITSTATUS(st) != 0) return 1;
  return 0;
}

size_t samgraph_num_epoch(void) { return Engine::Get().NumEpoch(); }
size_t samgraph_steps_per_epoch(void) { return Engine::Get().NumStep(); }
size_t samgraph_num_local_step(void) { return Engine::Get().NumLocalStep(); }
size_t samgraph_num_class(void) { return Engine::Get().ds.num_class; }
size_t samgraph_feat_dim(void) { return Engine::Get().ds.feat_dim; }
uint64_t samgraph_get_next_batch(void) { return Engine::Get().GetNextBatch(); }
void samgraph_sample_once(void) { Engine::Get().RunSampleOnce(); }

size_t samgraph_get_graph_num_src(uint64_t key, int g) { return Engine::Get().Current(key)->counts[3 * g + 1]; }
size_t samgraph_get_graph_num_dst(uint64_t key, int g) { return Engine::Get().Current(key)->counts[3 * g + 2]; }
size_t samgraph_get_graph_num_edge(uint64_t key, int g) { return Engine::Get().Current(key)->counts[3 * g + 0]; }

void samgraph_log_step(uint64_t e, uint64_t s, int item, double v) { auto &E = Engine::Get(); E.prof.LogStep(E.BatchKey(e, s), item, v); }
void samgraph_log_step_by_key(uint64_t key, int item, double v) { Engine::Get().prof.LogStep(key, item, v); }
void samgraph_log_step_add(uint64_t e, uint64_t s, int item, double v) { auto &E = Engine::Get(); E.prof.LogStepAdd(E.BatchKey(e, s), item, v); }
void samgraph_log_epoch_add(uint64_t e, int item, double v) { auto &E = Engine::Get(); E.prof.LogEpochAdd(E.BatchKey(e, 0), item, v); }
double samgraph_get_log_init_value(int item) { return (item >= 0 && item < sam::Profiler::kMaxInit) ? Engine::Get().prof.GetInit(item) : 0.0; }
double samgraph_get_log_step_value(uint64_t e, uint64_t s, int item) { auto &E = Engine::Get(); return E.prof.GetStep(E.BatchKey(e, s), item); }
double samgraph_get_log_step_value_by_key(uint64_t key, int item) { return Engine::Get().prof.GetStep(key, item); }
double samgraph_get_log_epoch_value(uint64_t e, int item) { return Engine::Get().prof.GetEpoch(e, item); }
void samgraph_report_init(void) {
  auto &p = Engine::Get().prof;
  std::printf("[Init] load dataset %.4f s | build cache %.4f s\n", p.GetInit(6), p.GetInit(10));
  std::fflush(stdout);
}
void samgraph_report_step(uint64_t e, uint64_t s) { Engine::Get().prof.ReportStep(e, s); }
void samgraph_report_step_average(uint64_t e, uint64_t s) { Engine::Get().prof.ReportStep(e, s); std::fflush(stdout); }
void samgraph_report_epoch(uint64_t e) { Engine::Get().prof.ReportEpoch(e); }
void samgraph_report_epoch_average(uint64_t e) { Engine::Get().prof.ReportEpoch(e); std::fflush(stdout); }
void samgraph_report_node_access(void) { Engine::Get().ReportNodeAccess(); } // operation.cc:472-480
void samgraph_trace_step_begin(uint64_t key, int item, uint64_t ts) { Engine::Get().prof.Trace(key, item, ts, true); }
void samgraph_trace_step_end(uint64_t key, int item, uint64_t ts) { Engine::Get().prof.Trace(key, item, ts, false); }
void samgraph_trace_step_begin_now(uint64_t key, int item) { Engine::Get().prof.Trace(key, item, now_us(), true); }
void samgraph_trace_step_end_now(uint64_t key, int item) { Engine::Get().prof.Trace(key, item, now_us(), false); }
void samgraph_dump_trace(void) { Engine::Get().prof.DumpTrace(); }

// ---- tensor hand-off (adapter.cc:62-193) ------------------------------------------------------
void samgraph_get_graph_feat(uint64_t key, samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  auto *b = E.Current(key);
  fill(out, b->feat, (int64_t)b->num_input, (int64_t)E.ds.feat_dim, 2, E.ds.feat_dtype, E.batch_device_type(), E.trainer_device());
}
void samgraph_get_graph_label(uint64_t key, samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  auto *b = E.Current(key);
  fill(out, b->label, (int64_t)b->num_seeds, 1, 1, GGMS_I64, E.batch_device_type(), E.trainer_device());
}
void samgraph_get_graph_row(uint64_t key, int l, samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  auto *b = E.Current(key);
  fill(out, b->row[l], (int64_t)b->counts[3 * l], 1, 1, GGMS_I32, E.batch_device_type(), E.trainer_device());
}
void samgraph_get_graph_col(uint64_t key, int l, samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  auto *b = E.Current(key);
  fill(out, b->col[l], (int64_t)b->counts[3 * l], 1, 1, GGMS_I32, E.batch_device_type(), E.trainer_device());
}
void samgraph_get_graph_data(uint64_t key, int l, samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  auto *b = E.Current(key);
  fill(out, b->data[l], b->data[l] ? (int64_t)b->counts[3 * l] : 0, 1, 1, GGMS_I32, E.batch_device_type(), E.trainer_device());
}
void samgraph_get_dataset_feat(samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  fill(out, E.ds.feat.ptr, (int64_t)E.ds.feat_rows, (int64_t)E.ds.feat_dim, 2, E.ds.feat_dtype, 0, 0);
}
void samgraph_get_dataset_label(samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  fill(out, E.ds.label.ptr, (int64_t)E.ds.num_node, 1, 1, GGMS_I64, 0, 0);
}
void samgraph_get_graph_input_nodes(uint64_t key, samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  auto *b = E.Current(key);
  fill(out, b->input_nodes, (int64_t)b->num_input, 1, 1, GGMS_I32, E.batch_device_type(), E.trainer_device());
}
void samgraph_get_graph_output_nodes(uint64_t key, samgraph_tensor_t *out) {
  auto &E = Engine::Get();
  auto *b = E.Current(key);
  fill(out, b->output_nodes, (int64_t)b->num_seeds, 1, 1, GGMS_I32, E.batch_device_type(), E.trainer_device());
}
void samgraph_batch_retain(uint64_t key) { Engine::Get().Retain(key); }
void samgraph_batch_release(uint64_t key) { Engine::Get().Release(key); }

} // extern "C"
