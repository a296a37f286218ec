// engine.h -- host orchestration behind the samgraph_* ABI (include/samgraph.h).
//
// Counterpart of the reference's Engine / GPUEngine / DistEngine
// (samgraph/common/engine.{h,cc}, cuda/cuda_engine.cc, dist/dist_engine.cc) for the
// deployments on the hot path: arch1 (one process, one GPU) and arch6 (one process
// per GPU, GGMS shards).  It owns device memory (hipMalloc, no framework allocator),
// streams, the shuffler, the sampler state and the feature cache, and drives the
// leaf operators of include/ggms.h.  Everything below is plain C++17 + HIP runtime.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/ggms.h"
#include "../../../include/samgraph.h"

namespace sam {

// ---- fatal checks: print + abort like LOG(FATAL)/CHECK (logging.cc:69-73) ----
[[noreturn]] void fatal(const char *file, int line, const std::string &msg);
#define SAM_CHECK(cond, msg)                                   \
  do {                                                         \
    if (!(cond)) ::sam::fatal(__FILE__, __LINE__, std::string("Check failed: " #cond " ") + (msg)); \
  } while (0)
#define SAM_HIP(call)                                          \
  do {                                                         \
    hipError_t e_ = (call);                                    \
    if (e_ != hipSuccess) ::sam::fatal(__FILE__, __LINE__, std::string(#call " -> ") + hipGetErrorString(e_)); \
  } while (0)
#define SAM_GGMS(call)                                         \
  do {                                                         \
    int rc_ = (call);                                          \
    if (rc_ != 0) ::sam::fatal(__FILE__, __LINE__, std::string(#call " -> ") + ggms_last_error()); \
  } while (0)
void log_info(const std::string &msg);

// ---- RunConfig: run_config.h + operation.cc:64-326 -----------------------------
enum Arch { kArch0 = 0, kArch1, kArch2, kArch3, kArch4, kArch5, kArch6, kArch7 };
struct RunConfig {
  std::unordered_map<std::string, std::string> raw;
  std::string dataset_path;
  int arch = kArch1;
  int sample_type = GGMS_KHOP3;
  size_t batch_size = 8000, num_epoch = 1;
  int cache_policy = 0;
  double cache_percentage = 0.0;
  size_t num_layer = 0;
  std::vector<size_t> fanout;
  size_t random_walk_length = 0, num_random_walk = 0, num_neighbor = 0;
  double random_walk_restart_prob = 0.0;
  size_t num_worker = 1;
  int sampler_device = 0, trainer_device = 0;
  bool trainer_on_host = false; // arch0 with trainer_ctx = cpu:N (host-only plumbing runs)
  size_t omp_thread_num = 1;    // arch0: threads of the sampling / extract loops (RunConfig::omp_thread_num)
  bool use_dist_graph = false;
  double dist_graph_percentage = 0.0;
  bool part_cache = false, gpu_extract = false;
  double replicate_percentage = 0.0; // of the cached slots, hottest first: kept on every GPU (hybrid store)
  bool configured = false;
  // extensions (optional keys)
  bool has_seed = false;
  uint64_t seed = 0;
  bool direct_table = true;
  size_t lookahead = 2; // batches sample_once() keeps enqueued beyond the one it was asked for (config key `lookahead`)
  size_t extract_streams = 2; // lean batches' gathers alternate between this many streams (1 or 2; config key `extract_streams`)
  size_t pipelines = 1; // batches the sampler itself has in flight: own stream + dedup table + workspace each (`pipelines`);
                        // measured on papers100M-shaped GCN: a second pipeline loses 2-3 % (the step is bound by the memory
                        // fabric, not by sampler latency) and costs a second 8 B x num_node table
  size_t presample_epoch = 0;
  size_t staged_serial_epochs = 0; // host-staged path: the first N epochs run the reference's serial, per-phase-timed sequence
  size_t staged_serial_steps = 0;  // ... or this worker's first N batches
  bool UsePresample() const { return UseGPUCache() && (cache_policy == 2 /*kCacheByPreSample*/); }
  bool UseGPUCache() const { return cache_percentage > 0 && arch != kArch1; } // run_config.h:124-126
};

// ---- Dataset: common.h:216-243 + engine.cc:109-443 -----------------------------
struct HostArray {
  void *ptr = nullptr;
  size_t bytes = 0;
  bool mapped_file = false, shared_anon = false, owned = false;
};
struct Dataset {
  size_t num_node = 0, num_edge = 0, num_class = 0, feat_dim = 0;
  int feat_dtype = GGMS_F32;
  size_t num_train = 0, num_valid = 0, num_test = 0;
  HostArray indptr, indices, feat, label, train_set, valid_set, test_set, ranking_nodes, prob_table, alias_table;
  bool feat_is_fake = false;
  bool feat_is_zero = false; // the stand-in table of a dataset without feat.bin that nobody has written: every row is zero
  // SAMGRAPH_EMPTY_FEAT = k (engine.cc:198-235): the feature table is a 2^k-row stand-in, row of node v = v & mask
  uint32_t feat_mask = 0xffffffffu;
  size_t feat_rows = 0; // rows of ds.feat (num_node, or 2^k)
};

// ---- Profiler log store: profiler.h:166-215 ------------------------------------
class Profiler {
 public:
  void Resize(size_t num_epoch, size_t num_step);
  void LogInit(int item, double v) { init_[item] = v; }
  void LogInitAdd(int item, double v) { init_[item] += v; }
  void LogStep(uint64_t key, int item, double v);
  void LogStepAdd(uint64_t key, int item, double v);
  void LogEpochAdd(uint64_t key, int item, double v);
  double GetInit(int item) const { return init_[item]; }
  double GetStep(uint64_t key, int item) const;
  double GetEpoch(uint64_t epoch, int item) const;
  void Trace(uint64_t key, int item, uint64_t ts, bool begin);
  void DumpTrace();
  void ReportStep(uint64_t epoch, uint64_t step);
  void ReportEpoch(uint64_t epoch);
  static constexpr int kMaxInit = 64, kMaxStep = 96, kMaxEpoch = 32;
  size_t num_step_ = 1, num_epoch_ = 1;

 private:
  double init_[kMaxInit] = {0};
  std::vector<double> step_;  // [key][item]
  std::vector<double> epoch_; // [epoch][item]
  struct TraceRec { uint64_t key; int item; uint64_t begin, end; };
  std::vector<TraceRec> trace_;
  std::mutex mu_;
};

// ---- one mini-batch in flight (Task / TrainGraph, common.h:246-283) -------------
struct Batch {
  uint64_t key = 0;
  int slot = -1;
  std::atomic<int> refs{0};
  bool in_use = false;
  bool host = false; // arch0 with a host trainer: the buffers below are host memory, complete when enqueued
  // device buffers, allocated once at their upper bounds
  std::vector<uint32_t *> row, col, data;
  uint32_t *input_nodes = nullptr, *output_nodes = nullptr;
  void *feat = nullptr;
  int64_t *label = nullptr;
  uint64_t *counts_dev = nullptr; // 3L+8: counts, status, then rows per tier {host miss, remote, local, replica}
  // pinned host copy (hipHostMalloc), valid after Finish()
  uint64_t *counts = nullptr;
  size_t num_seeds = 0, num_input = 0;
  uint64_t num_miss = 0;
  hipEvent_t ev_seeds = nullptr, ev_start = nullptr, ev_sampled = nullptr, ev_xstart = nullptr, ev_done = nullptr;
  // arch6 with `gpu_extract` off: the host-staged miss path (dist_loops.cc:1015-1207)
  uint32_t *miss_src = nullptr, *miss_dst = nullptr, *hit_src = nullptr, *hit_dst = nullptr;
  void *idx_ws = nullptr, *miss_rows_dev = nullptr;
  void *miss_rows_host = nullptr;   // hipHostMalloc (pinned): CPU-gathered miss rows, copied down chunk by chunk
  uint32_t *miss_ids_host = nullptr; // hipHostMalloc
  hipEvent_t ev_ids = nullptr;       // the miss ids have reached the host
  hipEvent_t ev_label = nullptr;     // the labels of the batch are gathered
  // the feature gather's own start / end timestamps, riding on its dispatch packet (include/ggms.h, launch timer);
  // lean: this batch's extract stream carried the gather and nothing else (EnqueueOne)
  ggms_launch_timer_t *gather_timer = nullptr;
  bool lean = false;
};

class Engine {
 public:
  static Engine &Get();
  Engine();
  ~Engine();
  RunConfig cfg;
  Dataset ds;
  Profiler prof;

  void Configure(const std::unordered_map<std::string, std::string> &kv);
  void DataInit();                                    // load dataset (host only; fork-safe)
  void SampleInit(int worker_id, const std::string &ctx); // device state for sampling
  void TrainInit(int worker_id, const std::string &ctx);  // feature cache / extract state
  void Init();                                        // arch1: all three
  void Start();
  void Shutdown();
  void RunSampleOnce(bool background = false);
  bool EnqueueOne(bool background);
  uint64_t GetNextBatch();
  void ExtractStart(int count);
  Batch *Current(uint64_t key);
  void Retain(uint64_t key);
  void Release(uint64_t key);

  size_t NumEpoch() const { return cfg.num_epoch; }
  size_t NumStep() const { return num_global_step_; }
  size_t NumLocalStep() const { return num_local_step_; }
  uint64_t BatchKey(uint64_t epoch, uint64_t step) const { return epoch * num_global_step_ + step; }
  int trainer_device() const { return device_; }
  int batch_device_type() const { return (cfg.arch == kArch0 && cfg.trainer_on_host) ? 0 : 2; } // DeviceType, common.h:48
  void Barrier(const char *what = "step");
  void *OpenPeer(const hipIpcMemHandle_t &handle, uint32_t peer, size_t bytes, const char *what);

 private:
  // dataset
  void LoadDataset();
  HostArray MapFile(const std::string &name, size_t bytes, bool to_shared_anon);
  // shuffler (cuda/cuda_shuffler.cc, dist/dist_shuffler_aligned.cc)
  void ShufflerInit();
  bool ShufflerNext(Batch *b, hipStream_t copy_stream); // false at end of training
  void Reshuffle();
  // GGMS
  void DetectTopo(); // PartitionSolver::DetectTopo (dist_graph.cu:684-726): P2P reachability + link rates, probed in a forked child
  ggms_topology_t topo_{};
  bool topo_valid_ = false;
  void UploadGraph();
  void Presample();
  void BuildCache();
  Batch *AcquireSlot(bool background);
  // `gpu_extract` off (SGNN mode of arch6): miss ids -> host, CPU gather into pinned memory, async H2D, combine
  // (a partial cache: its table exists; no cache at all: every row is a miss.  A FULL cache has no table and no misses.)
  bool StagedHostTier() const { return cfg.arch == kArch6 && !cfg.gpu_extract && (cache_table_ != nullptr || !cfg.UseGPUCache()); }
  void StagedExtract(Batch *b, hipStream_t ss, hipStream_t xs);
  void HostGatherRows(char *rows, const uint32_t *ids, size_t first, size_t count); // ExtractMissData on the host team
  std::unique_ptr<class Team> host_team_;
  size_t staged_batches_ = 0;
  void SanityCheckBatch(const uint32_t *seeds, size_t n); // SAMGRAPH_SANITY_CHECK
  std::vector<bool> sanity_seen_;
  uint32_t *node_access_dev_ = nullptr; // SAMGRAPH_LOG_NODE_ACCESS[_SIMPLE]: visits per node (input nodes of every batch)
 public:
  void ReportNodeAccess();
 private:
  // arch0 (cpu_engine.cc): sampler + extractor on the host cores
  struct CpuPath;
  std::unique_ptr<CpuPath> cpu_;
  void CpuInit();
  bool CpuEnqueueOne(bool background);
  void CpuShutdown();
  void Finish(Batch *b, Batch *prev);

  bool data_ready_ = false, sample_ready_ = false, train_ready_ = false, shutdown_ = false;
  int worker_id_ = 0, device_ = 0;
  hipStream_t stream_ = nullptr;         // shuffle + sampling (latency-bound)
  hipStream_t stream_extract_ = nullptr; // feature gather (HBM-bound): overlaps the next batch's sampling
  // a second one: consecutive lean batches' gathers alternate and may overlap -- no wait packet between two gathers, and
  // one gather's head fills the other's tail (default workload -5 %, profiles/r05_ab_extract_streams.txt)
  hipStream_t stream_extract2_ = nullptr;
  // device graph
  uint32_t *d_indptr_ = nullptr, *d_indices_ = nullptr;
  std::vector<void *> part_indptr_, part_indices_; // P+1 entries (slot P = host CSR)
  ggms_graph_t graph_{};
  // sampler state
  ggms_hashtable_t ht_{};
  ggms_sample_extra_t extra_{};
  void *d_prob_ = nullptr, *d_alias_ = nullptr;
  void *states_ = nullptr;
  size_t num_states_ = 0;
  void *ws_ = nullptr;
  size_t ws_bytes_ = 0;
  // sampling pipelines 1..K-1 (pipeline 0 is {stream_, ht_, ws_}); the RNG pool is consumed in batch order
  struct Pipe {
    hipStream_t stream = nullptr;
    ggms_hashtable_t ht{};
    void *ws = nullptr;
    hipEvent_t rng_done = nullptr;
  };
  std::vector<Pipe> pipes_;
  size_t enq_count_ = 0;
  hipEvent_t last_rng_done_ = nullptr;
  size_t max_seeds_ = 0, max_unique_ = 0;
  std::vector<size_t> max_input_, max_edges_;
  // shuffler
  std::vector<uint32_t> shuf_host_;
  uint32_t *shuf_dev_ = nullptr;
  size_t num_data_ = 0, num_local_data_ = 0, num_local_step_ = 0, num_global_step_ = 0;
  size_t global_step_offset_ = 0, global_data_offset_ = 0;
  size_t cur_epoch_ = 0, cur_step_ = 0;
  bool shuf_initialized_ = false;
  // ggms_sample_extra_t.seeds_distinct: the train set holds no node twice (checked once), so a batch's seeds are
  // distinct unless both copies of a node that pads the aligned epoch (dist_shuffler_aligned.cc:52-54) fall into it
  bool train_distinct_ = false;
  std::vector<std::pair<size_t, size_t>> pad_pairs_; // this worker's slice, this epoch: local positions of such copies
  bool BatchSeedsDistinct(size_t offset, size_t size) const;
  // features
  uint32_t *cache_table_ = nullptr;            // id -> slot (GPUCacheManager::_sampler_gpu_hashtable)
  std::vector<void *> cache_parts_;            // shard base pointers (local or IPC-mapped)
  uint32_t num_cache_part_ = 0;
  size_t num_cached_nodes_ = 0;
  size_t num_replica_ = 0;                     // hybrid store: slots [0, num_replica_) live in d_replica_ on every GPU
  void *d_replica_ = nullptr;
  void *d_feat_ = nullptr;                     // full feature table on the device (arch1) ...
  const void *feat_src_ = nullptr;             // ... or device-mapped host memory (gpu_extract / miss tier)
  const void *label_src_ = nullptr;
  // batches
  std::vector<std::unique_ptr<Batch>> slots_;
  std::deque<Batch *> pool_;
  std::mutex pool_mu_;
  std::condition_variable pool_cv_;
  Batch *current_ = nullptr;
  std::thread bg_;
  size_t fg_calls_ = 0, fg_enqueued_ = 0; // foreground sample_once() calls / batches enqueued for them
  std::atomic<bool> bg_stop_{false};
  std::atomic<bool> bg_running_{false};
  // shared (arch6): control block inherited through fork
  struct Shared;
  Shared *shared_ = nullptr;
};

} // namespace sam
