// samgraph_no_train -- pure C++ caller of the samgraph_* C ABI (include/samgraph.h): sample + extract loop
// without PyTorch/DGL, the role samgraph/main.cc (samgraph_cpp_no_train) plays in the reference.
//   build: make -C xgnn_amd/csrc driver      run: build/samgraph_no_train --dataset-path DIR [--key value ...]
// Every "--some-key v" becomes config key "some_key" = v; defaults follow example/samgraph/common_config.py.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "samgraph.h"

int main(int argc, char **argv) {
  std::map<std::string, std::string> cfg = {
      {"_arch", "1"}, {"_sample_type", "7"}, {"batch_size", "8000"}, {"num_epoch", "3"}, {"_cache_policy", "0"},
      {"cache_percentage", "0"}, {"max_sampling_jobs", "10"}, {"max_copying_jobs", "1"}, {"omp_thread_num", "40"},
      {"num_layer", "2"}, {"num_hidden", "256"}, {"lr", "0.003"}, {"dropout", "0.5"}, {"num_fanout", "2"},
      {"fanout", "25 10"}, {"sampler_ctx", "cuda:0"}, {"trainer_ctx", "cuda:0"}, {"dataset_path", ""}};
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string k = argv[i];
    if (k.rfind("--", 0) != 0) { std::fprintf(stderr, "expected --key value, got %s\n", argv[i]); return 2; }
    k = k.substr(2);
    for (auto &c : k) if (c == '-') c = '_';
    cfg[k] = argv[i + 1];
  }
  if (cfg["dataset_path"].empty()) { std::fprintf(stderr, "--dataset-path is required\n"); return 2; }
  // "fanout" given as "25 10": derive the counts like the Python front-end does (common_config.py)
  size_t nf = 0;
  { bool in = false; for (char c : cfg["fanout"]) { if (c != ' ' && !in) { in = true; ++nf; } else if (c == ' ') in = false; } }
  cfg["num_fanout"] = std::to_string(nf);
  cfg["num_layer"] = std::to_string(nf);
  std::vector<const char *> keys, vals;
  for (auto &kv : cfg) { keys.push_back(kv.first.c_str()); vals.push_back(kv.second.c_str()); }
  samgraph_config(keys.data(), vals.data(), keys.size());
  samgraph_init();
  const size_t num_epoch = samgraph_num_epoch(), steps = samgraph_steps_per_epoch();
  for (size_t e = 0; e < num_epoch; ++e) {
    auto t0 = std::chrono::steady_clock::now();
    for (size_t b = 0; b < steps; ++b) {
      samgraph_sample_once();
      samgraph_get_next_batch();
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // item codes: kLogEpochSampleTime = 0, kLogEpochCopyTime = 8, kLogEpochFeatureBytes = 12, kLogEpochNumSample = 15
    const double ts = samgraph_get_log_epoch_value(e, 0), tc = samgraph_get_log_epoch_value(e, 8);
    const double fb = samgraph_get_log_epoch_value(e, 12), ns = samgraph_get_log_epoch_value(e, 15);
    std::printf("[epoch %zu] %zu steps, wall %.4f s | %.0f edges, sample %.6f s -> %.3f M SEPS | extract %.6f s -> %.2f GB/s\n",
                e, steps, wall, ns, ts, ns / ts / 1e6, tc, fb / tc / 1e9);
  }
  samgraph_report_init();
  samgraph_shutdown();
  return 0;
}
