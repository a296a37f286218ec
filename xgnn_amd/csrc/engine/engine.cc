// engine.cc -- host orchestration (see engine.h).  Citations are relative to
// /root/reference/samgraph/common/.
#include "engine.h"
#include "team.h"

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <thread>

namespace sam {

void fatal(const char *file, int line, const std::string &msg) {
  std::fprintf(stderr, "[samgraph-amd FATAL] %s:%d: %s\n", file, line, msg.c_str());
  std::fflush(stderr);
  std::abort(); // logging.cc:69-73: failed CHECK aborts the process
}

void log_info(const std::string &msg) {
  static const bool on = [] { const char *e = getenv("SAMGRAPH_LOG_LEVEL"); return e && (std::string(e) == "info" || std::string(e) == "debug"); }();
  if (on) std::fprintf(stderr, "[samgraph-amd] %s\n", msg.c_str());
}

// process-shared control block for arch6 workers (dist_graph.cu:566-636: anonymous MAP_SHARED mmap +
// PTHREAD_PROCESS_SHARED barrier, inherited through fork)
struct Engine::Shared {
  // worker barrier with a deadline (pthread_barrier_wait has none): arrivals of the current generation + the
  // generation counter; zero-initialised by the anonymous mapping, lock-free atomics are valid across processes
  std::atomic<uint32_t> arrived, generation;
  int num_worker;
  static constexpr int kMaxWorker = 16;
  hipIpcMemHandle_t graph_indptr[kMaxWorker], graph_indices[kMaxWorker], feat_part[kMaxWorker];
  size_t indptr_words[kMaxWorker], indices_words[kMaxWorker], feat_rows[kMaxWorker];
};

Engine &Engine::Get() {
  static Engine e;
  return e;
}

static int parse_device(const std::string &ctx) {
  // "cuda:3" / "cpu:0" (Context(std::string), common.cc)
  auto p = ctx.find(':');
  int id = p == std::string::npos ? 0 : std::atoi(ctx.c_str() + p + 1);
  const char *force = getenv("SAMGRAPH_FORCE_DEVICE"); // test hook: all workers on one physical GPU
  if (force) id = std::atoi(force);
  return id;
}

// ------------------------------------------------------------------ configuration
void Engine::Configure(const std::unordered_map<std::string, std::string> &kv_in) {
  auto kv = kv_in;
  SAM_CHECK(!cfg.configured, "samgraph_config called twice");
  // required keys, operation.cc:68-81
  for (const char *k : {"dataset_path", "_arch", "_sample_type", "batch_size", "num_epoch", "_cache_policy",
                        "cache_percentage", "max_sampling_jobs", "max_copying_jobs", "omp_thread_num", "num_layer",
                        "num_hidden", "lr", "dropout"})
    SAM_CHECK(kv.count(k), std::string("missing config key ") + k);
  cfg.raw = kv;
  cfg.dataset_path = kv["dataset_path"];
  cfg.arch = std::stoi(kv["_arch"]);
  cfg.sample_type = std::stoi(kv["_sample_type"]);
  cfg.batch_size = std::stoull(kv["batch_size"]);
  cfg.num_epoch = std::stoull(kv["num_epoch"]);
  cfg.cache_policy = std::stoi(kv["_cache_policy"]);
  cfg.cache_percentage = std::stod(kv["cache_percentage"]);
  cfg.num_layer = std::stoull(kv["num_layer"]);
  SAM_CHECK(cfg.batch_size > 0 && cfg.num_layer >= 1 && cfg.num_layer <= 16, "batch_size >= 1 and 1 <= num_layer <= 16");
  // every deployment: arch0's sampler / extractor team and arch6's host-staged miss path (gpu_extract off) use it
  cfg.omp_thread_num = std::max<size_t>(1, std::stoull(kv["omp_thread_num"]));
  switch (cfg.arch) { // operation.cc:101-148
    case kArch1:
      SAM_CHECK(kv.count("sampler_ctx") && kv.count("trainer_ctx"), "arch1 needs sampler_ctx/trainer_ctx");
      cfg.sampler_device = parse_device(kv["sampler_ctx"]);
      cfg.trainer_device = parse_device(kv["trainer_ctx"]);
      cfg.num_worker = 1;
      break;
    case kArch0: // CPU sampler + extractor, trainer on a GPU or (plumbing runs) on the host
      SAM_CHECK(kv.count("sampler_ctx") && kv.count("trainer_ctx"), "arch0 needs sampler_ctx/trainer_ctx");
      cfg.trainer_on_host = kv["trainer_ctx"].rfind("cpu", 0) == 0;
      cfg.trainer_device = parse_device(kv["trainer_ctx"]);
      cfg.num_worker = 1;
      break;
    case kArch6:
      SAM_CHECK(kv.count("num_worker"), "arch6 needs num_worker");
      cfg.num_worker = std::stoull(kv["num_worker"]);
      SAM_CHECK(cfg.num_worker >= 1, "num_worker >= 1");
      break;
    default:
      fatal(__FILE__, __LINE__, "only arch0 (CPU), arch1 (standalone) and arch6 (SGNN/XGNN) are built; see DESIGN.md");
  }
  if (cfg.sample_type != GGMS_RANDOM_WALK) { // operation.cc:150-163
    SAM_CHECK(kv.count("num_fanout") && kv.count("fanout"), "khop sampling needs num_fanout/fanout");
    size_t nf = std::stoull(kv["num_fanout"]);
    std::stringstream ss(kv["fanout"]);
    for (size_t i = 0; i < nf; ++i) {
      size_t f = 0;
      ss >> f;
      SAM_CHECK(f > 0, "fanout: num_fanout positive integers expected");
      cfg.fanout.push_back(f);
    }
  } else { // :164-175
    cfg.random_walk_length = std::stoull(kv["random_walk_length"]);
    cfg.random_walk_restart_prob = std::stod(kv["random_walk_restart_prob"]);
    cfg.num_random_walk = std::stoull(kv["num_random_walk"]);
    cfg.num_neighbor = std::stoull(kv["num_neighbor"]);
    SAM_CHECK(cfg.random_walk_length > 0 && cfg.num_random_walk > 0 && cfg.num_neighbor > 0,
              "random walk needs random_walk_length, num_random_walk and num_neighbor >= 1");
    cfg.fanout.assign(cfg.num_layer, cfg.num_neighbor);
  }
  if (kv.count("use_dist_graph")) { // :191-203
    cfg.dist_graph_percentage = std::stod(kv["use_dist_graph"]);
    cfg.use_dist_graph = cfg.dist_graph_percentage > 0.0;
  }
  if (kv.count("part_cache") && kv["part_cache"] == "True") { // :205-211
    SAM_CHECK(cfg.arch == kArch6, "partition cache can only be used in arch6");
    cfg.part_cache = true;
  }
  if (kv.count("gpu_extract") && kv["gpu_extract"] == "True") cfg.gpu_extract = (cfg.arch == kArch6); // :229-235
  // extension: the hottest fraction of the CACHED slots is kept on every GPU instead of being sharded (what the
  // reference's PartitionSolver buys with replica placement on NVLink, dist_graph.cu:40-222); 0 = pure modulo shards
  if (kv.count("replicate_percentage")) cfg.replicate_percentage = std::stod(kv["replicate_percentage"]);
  if (kv.count("presample_epoch")) cfg.presample_epoch = std::stoull(kv["presample_epoch"]); // operation.cc:184-189
  if (kv.count("seed")) { cfg.has_seed = true; cfg.seed = std::stoull(kv["seed"]); }
  if (kv.count("hash_table")) cfg.direct_table = kv["hash_table"] != "hashed";
  if (kv.count("lookahead")) cfg.lookahead = std::stoull(kv["lookahead"]);
  if (kv.count("staged_serial_epochs")) cfg.staged_serial_epochs = std::stoull(kv["staged_serial_epochs"]);
  if (kv.count("staged_serial_steps")) cfg.staged_serial_steps = std::stoull(kv["staged_serial_steps"]);
  if (kv.count("extract_streams")) cfg.extract_streams = std::max<size_t>(1, std::min<size_t>(2, std::stoull(kv["extract_streams"])));
  if (const char *e = getenv("SAMGRAPH_EXTRACT_STREAMS")) cfg.extract_streams = (e[0] == '1') ? 1 : 2; // A/B hook
  if (kv.count("pipelines")) cfg.pipelines = std::max<size_t>(1, std::min<size_t>(4, std::stoull(kv["pipelines"])));
  if (cfg.lookahead + 1 < cfg.pipelines) cfg.pipelines = cfg.lookahead + 1; // nothing to overlap without batches ahead
  SAM_CHECK(cfg.sample_type >= GGMS_KHOP0 && cfg.sample_type <= GGMS_KHOP3, "unknown sample type");
  if (cfg.sample_type == GGMS_WEIGHTED_KHOP || cfg.sample_type == GGMS_KHOP2 || cfg.sample_type == GGMS_KHOP1 ||
      cfg.sample_type == GGMS_WEIGHTED_KHOP_PREFIX ||
      cfg.sample_type == GGMS_WEIGHTED_KHOP_HASH_DEDUP) // dist_loops.cc:167-168,171-172,209-210,219-220,227-228
    SAM_CHECK(!cfg.use_dist_graph, "this algorithm not support DistGraph engine");
  // shard base pointers travel in the kernels' arguments (include/ggms.h, GGMS_MAX_PARTS): a larger group is refused
  // HERE, before any shard is built, exported or mapped (the reference's device pointer tables take any num_part)
  if (cfg.arch == kArch6 && (cfg.use_dist_graph || cfg.part_cache))
    SAM_CHECK(cfg.num_worker <= GGMS_MAX_PARTS, "use_dist_graph / part_cache: at most GGMS_MAX_PARTS (8) workers share a sharded store");
  cfg.configured = true;
}

// ------------------------------------------------------------------ dataset
HostArray Engine::MapFile(const std::string &name, size_t bytes, bool to_shared_anon) {
  HostArray a;
  a.bytes = bytes;
  const std::string path = cfg.dataset_path + name;
  int fd = open(path.c_str(), O_RDONLY);
  SAM_CHECK(fd >= 0, "cannot open " + path);
  struct stat st;
  fstat(fd, &st);
  SAM_CHECK((size_t)st.st_size >= bytes, path + " is smaller than meta.txt says");
  void *m = bytes ? mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr; // Tensor::FromMmap
  SAM_CHECK(bytes == 0 || m != MAP_FAILED, "mmap failed for " + path);
  close(fd);
  if (to_shared_anon && bytes) {
    // ConverToAnonMmap, engine.cc:91-107: workers forked later share one locked copy
    void *s = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    SAM_CHECK(s != MAP_FAILED, "anonymous shared mmap failed");
    std::memcpy(s, m, bytes);
    munmap(m, bytes);
    a.ptr = s;
    a.shared_anon = true;
  } else {
    a.ptr = m;
    a.mapped_file = true;
  }
  return a;
}

static bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

void Engine::LoadDataset() {
  if (cfg.dataset_path.back() != '/') cfg.dataset_path.push_back('/');
  // meta.txt, engine.cc:126-142; names constant.cc:23-51
  std::ifstream meta_file(cfg.dataset_path + "meta.txt");
  SAM_CHECK(meta_file.good(), "cannot read " + cfg.dataset_path + "meta.txt");
  std::unordered_map<std::string, size_t> meta;
  std::string line;
  while (std::getline(meta_file, line)) {
    std::istringstream iss(line);
    std::string k, v;
    if (!(iss >> k >> v)) break;
    if (k == "FEAT_DATA_TYPE") {
      static const std::map<std::string, int> names = {{"F32", GGMS_F32}, {"F64", GGMS_F64}, {"F16", GGMS_F16},
                                                       {"U8", GGMS_U8},   {"I32", GGMS_I32}, {"I8", GGMS_I8},
                                                       {"I64", GGMS_I64}};
      SAM_CHECK(names.count(v), "unknown FEAT_DATA_TYPE " + v);
      ds.feat_dtype = names.at(v);
    } else {
      meta[k] = std::stoull(v);
    }
  }
  for (const char *k : {"NUM_NODE", "NUM_EDGE", "FEAT_DIM", "NUM_CLASS", "NUM_TRAIN_SET", "NUM_TEST_SET", "NUM_VALID_SET"})
    SAM_CHECK(meta.count(k), std::string("meta.txt lacks ") + k);
  ds.num_node = meta["NUM_NODE"]; ds.num_edge = meta["NUM_EDGE"]; ds.feat_dim = meta["FEAT_DIM"];
  ds.num_class = meta["NUM_CLASS"]; ds.num_train = meta["NUM_TRAIN_SET"]; ds.num_test = meta["NUM_TEST_SET"];
  ds.num_valid = meta["NUM_VALID_SET"];
  const bool share = cfg.arch == kArch6; // forked workers read the same pages
  ds.indptr = MapFile("indptr.bin", (ds.num_node + 1) * 4, share);
  ds.indices = MapFile("indices.bin", ds.num_edge * 4, share);
  // SAMGRAPH_FAKE_FEAT_DIM (run_config.cc:156-159, engine.cc:202-204): pretend the features have this width and
  // do not read feat.bin -- lets a big graph run without its feature file
  size_t fake_dim = 0;
  if (const char *e = getenv("SAMGRAPH_FAKE_FEAT_DIM")) fake_dim = std::strtoull(e, nullptr, 10);
  if (fake_dim) ds.feat_dim = fake_dim;
  const size_t row_bytes = ds.feat_dim * ggms_dtype_bytes(ds.feat_dtype);
  ds.feat_rows = ds.num_node;
  size_t empty_bits = 0; // SAMGRAPH_EMPTY_FEAT = k (run_config.cc:137-139, engine.cc:205-207): a 2^k-row stand-in table,
                         // node v reads row v & (2^k - 1) (gpu_mock_extract / cpu_mock_extract)
  if (const char *e = getenv("SAMGRAPH_EMPTY_FEAT")) empty_bits = std::strtoull(e, nullptr, 10);
  if (empty_bits) {
    SAM_CHECK(empty_bits < 32, "SAMGRAPH_EMPTY_FEAT out of range");
    ds.feat_rows = (size_t)1 << empty_bits;
    ds.feat_mask = (uint32_t)(ds.feat_rows - 1);
    ds.feat.bytes = ds.feat_rows * row_bytes;
    ds.feat.ptr = mmap(nullptr, ds.feat.bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    SAM_CHECK(ds.feat.ptr != MAP_FAILED, "feature mmap failed");
    ds.feat.shared_anon = true;
    ds.feat_is_fake = true;
    // the reference leaves the stand-in uninitialised; ours starts as the first 2^k rows of feat.bin where that file
    // exists (any content is as valid, and this one can be checked), zeros otherwise
    if (!fake_dim && file_exists(cfg.dataset_path + "feat.bin")) {
      HostArray f = MapFile("feat.bin", std::min(ds.feat_rows, ds.num_node) * row_bytes, false);
      std::memcpy(ds.feat.ptr, f.ptr, f.bytes);
    }
  } else if (!fake_dim && file_exists(cfg.dataset_path + "feat.bin")) {
    ds.feat = MapFile("feat.bin", ds.num_node * row_bytes, share);
  } else { // engine.cc:199-235: datasets without feat.bin get an (uninitialised) table; ours is zero-filled
    ds.feat.bytes = ds.num_node * row_bytes;
    ds.feat.ptr = mmap(nullptr, ds.feat.bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    SAM_CHECK(ds.feat.ptr != MAP_FAILED, "feature mmap failed");
    ds.feat.shared_anon = true;
    ds.feat_is_fake = true;
    // An anonymous mapping nobody has written is ONE zero page behind every address: a host gather of such a table reads
    // 4 KB over and over and measures the cache.  SAMGRAPH_FILL_FAKE_FEAT=1 gives every page a frame of its own (and the
    // rows a checkable content: 32-bit word w of the table holds w), filled by omp_thread_num threads.
    ds.feat_is_zero = !getenv("SAMGRAPH_FILL_FAKE_FEAT");
    if (!ds.feat_is_zero) {
      Team team((int)cfg.omp_thread_num);
      uint32_t *words = (uint32_t *)ds.feat.ptr;
      team.ParallelFor(ds.feat.bytes / 4, [&](size_t lo, size_t hi, int) {
        for (size_t w = lo; w < hi; ++w) words[w] = (uint32_t)w;
      });
    }
  }
  if (file_exists(cfg.dataset_path + "label.bin")) {
    ds.label = MapFile("label.bin", ds.num_node * 8, share);
  } else {
    ds.label.bytes = ds.num_node * 8;
    ds.label.ptr = mmap(nullptr, ds.label.bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    ds.label.shared_anon = true;
  }
  ds.train_set = MapFile("train_set.bin", ds.num_train * 4, false);
  ds.test_set = MapFile("test_set.bin", ds.num_test * 4, false);
  ds.valid_set = MapFile("valid_set.bin", ds.num_valid * 4, false);
  if (cfg.sample_type == GGMS_WEIGHTED_KHOP || cfg.sample_type == GGMS_WEIGHTED_KHOP_HASH_DEDUP) { // engine.cc:372-384
    ds.prob_table = MapFile("prob_table.bin", ds.num_edge * 4, false);
    ds.alias_table = MapFile("alias_table.bin", ds.num_edge * 4, false);
  }
  if (cfg.sample_type == GGMS_WEIGHTED_KHOP_PREFIX) // :373-378; the kernels see it through extra_.prob_table
    ds.prob_table = MapFile("prob_prefix_table.bin", ds.num_edge * 4, false);
  if (cfg.UseGPUCache()) { // engine.cc:395-440
    static const char *rank_files[] = {"cache_by_degree.bin", "cache_by_heuristic.bin", nullptr, "cache_by_degree_hop.bin",
                                       nullptr, "cache_by_fake_optimal.bin", nullptr, "cache_by_random.bin"};
    if (cfg.UsePresample()) {
      // filled by worker 0 in SampleInit, read by every worker (dist_engine.cc:455-466): shared pages
      ds.ranking_nodes.bytes = ds.num_node * 4;
      ds.ranking_nodes.ptr = mmap(nullptr, ds.ranking_nodes.bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
      SAM_CHECK(ds.ranking_nodes.ptr != MAP_FAILED, "ranking mmap failed");
      ds.ranking_nodes.shared_anon = true;
    } else {
      SAM_CHECK(cfg.cache_policy >= 0 && cfg.cache_policy < 8 && rank_files[cfg.cache_policy],
                "cache policy not built (presample_static / dynamic): see DESIGN.md");
      ds.ranking_nodes = MapFile(rank_files[cfg.cache_policy], ds.num_node * 4, false);
    }
  }
}

void Engine::DataInit() {
  SAM_CHECK(cfg.configured, "samgraph_config first");
  if (data_ready_) return;
  auto t0 = std::chrono::steady_clock::now();
  LoadDataset();
  if (cfg.arch == kArch6) {
    shared_ = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    SAM_CHECK(shared_ != MAP_FAILED, "control block mmap failed");
    SAM_CHECK(cfg.num_worker <= (size_t)Shared::kMaxWorker, "too many workers");
    static_assert(std::atomic<uint32_t>::is_always_lock_free, "the worker barrier lives in shared memory");
    shared_->arrived.store(0);
    shared_->generation.store(0);
    shared_->num_worker = (int)cfg.num_worker;
    // DistGraph::DistGraph -> PartitionSolver (dist_graph.cu:592-594): which GPUs reach which, before anything is placed
    if (cfg.num_worker > 1 && (cfg.use_dist_graph || cfg.part_cache)) DetectTopo();
  }
  prof.LogInit(/*kLogInitL2LoadDataset*/ 6, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  data_ready_ = true;
}

static double ipc_timeout_s();

// PartitionSolver::PartitionSolver + DetectTopo (dist_graph.cu:673-726): the parent must not touch the GPU before it
// forks its workers, so -- like the reference -- the probe runs in a forked child that sees every device: P2P
// reachability of every pair and a timed 128-MiB copy per reachable pair (ggms_detect_topology), kept in the
// reference's file format (read back here; reused by later runs on the same node, `SAMGRAPH_TOPO_FILE` names it).
// What the answer is used for: workers dereference each other's shards in place (DeviceDistGraph / DeviceDistFeature),
// so a pair of workers' GPUs that cannot reach each other is fatal HERE, with the pair named, instead of as a refused
// hipIpcOpenMemHandle after every shard has been built.  The reference's solver goes on to search clique placements
// for partially connected NVLink boxes (:728-777); an MI355X node is one clique (every GPU pair has its own xGMI
// link), so the placement is the modulo sharding of the whole group and the matrix is logged, not searched.
// A probe that cannot run (no device visible to the child, child killed) is a warning: placement does not depend on it.
void Engine::DetectTopo() {
  if (getenv("SAMGRAPH_FORCE_DEVICE")) return; // one-GPU rehearsal: every worker on one device, nothing to probe
  std::string file;
  if (const char *e = getenv("SAMGRAPH_TOPO_FILE")) {
    file = e;
  } else {
    const char *vis = getenv("HIP_VISIBLE_DEVICES");
    if (!vis) vis = getenv("ROCR_VISIBLE_DEVICES");
    std::string tag = vis ? vis : "all";
    for (auto &c : tag) if (!isalnum((unsigned char)c)) c = '_';
    const char *tmp = getenv("TMPDIR");
    file = std::string(tmp && *tmp ? tmp : "/tmp") + "/.detect_topo_amd_" + std::to_string((long)getuid()) + "_" + tag; // Constant::kDetectTopoFile
  }
  auto usable = [&](const ggms_topology_t &t) { return t.num_device >= (int)cfg.num_worker; };
  const auto t0 = std::chrono::steady_clock::now();
  bool probed = false;
  if (ggms_topology_read_host(&topo_, file.c_str()) != GGMS_OK || !usable(topo_)) {
    const pid_t pid = fork();
    SAM_CHECK(pid != -1, "fork of the topology probe failed");
    if (pid == 0) { // DetectTopo_child, :779-884
      ggms_topology_t t;
      int rc = ggms_detect_topology(&t, 0, 2);
      if (rc != GGMS_OK) fprintf(stderr, "[samgraph-amd] topology probe: %s\n", ggms_last_error());
      if (rc == GGMS_OK) rc = ggms_topology_write_host(&t, file.c_str(), "HIP_VISIBLE_DEVICES order");
      _exit(rc == GGMS_OK ? 0 : 1);
    }
    int wstatus = 0;
    bool done = false;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < std::min(120.0, ipc_timeout_s())) {
      const pid_t r = waitpid(pid, &wstatus, WNOHANG);
      if (r == pid || r == -1) { done = true; break; }
      usleep(20000);
    }
    if (!done) { // a probe that hangs must not hold the run
      kill(pid, SIGKILL);
      waitpid(pid, &wstatus, 0);
      wstatus = -1;
    }
    probed = done && WIFEXITED(wstatus) && WEXITSTATUS(wstatus) == 0;
    if (!probed || ggms_topology_read_host(&topo_, file.c_str()) != GGMS_OK) {
      fprintf(stderr, "[samgraph-amd] warning: the topology probe did not complete (%s); placing %zu modulo shards without it\n",
              done ? "child failed" : "child timed out", cfg.num_worker);
      return;
    }
  }
  SAM_CHECK(usable(topo_), "arch6 with " + std::to_string(cfg.num_worker) + " workers, but the node shows " +
                               std::to_string(topo_.num_device) + " GPUs (" + file + ")");
  topo_valid_ = true;
  for (size_t i = 0; i < cfg.num_worker; ++i)
    for (size_t j = 0; j < cfg.num_worker; ++j)
      if (i != j && !topo_.can_access[i][j])
        fatal(__FILE__, __LINE__, "GPU " + std::to_string(i) + " cannot access GPU " + std::to_string(j) +
                                      " (hipDeviceCanAccessPeer, " + file + "): workers read each other's GGMS shards in "
                                      "place (use_dist_graph / part_cache), which needs P2P access between every pair of "
                                      "their GPUs");
  std::ostringstream ss; // "Topology Detect Debug", :714-722
  ss << "topology (" << (probed ? "probed in " : "read from " + file + " in ")
     << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s): GB/s INTO row FROM column\n";
  char cell[32];
  for (size_t i = 0; i < cfg.num_worker; ++i) {
    for (size_t j = 0; j < cfg.num_worker; ++j) {
      snprintf(cell, sizeof(cell), "%8.1f ", topo_.copy_GBps[i][j]);
      ss << cell;
    }
    ss << "\n";
  }
  log_info(ss.str());
  prof.LogInit(/*kLogInitL3DistGraphDetectTopo: extension slot*/ 40, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
}

// Every wait on another worker has a deadline (SAMGRAPH_IPC_TIMEOUT_S, default 300 s): a worker that died, or a
// hipIpcOpenMemHandle that never returns (seen on ROCm 7.2 for exporter sizes with bit 31 set, include/ggms.h
// ggms_ipc_safe_bytes), must end the run with a message instead of holding it until somebody's time limit.
static double ipc_timeout_s() {
  static const double v = [] { const char *e = getenv("SAMGRAPH_IPC_TIMEOUT_S"); const double x = e ? atof(e) : 0; return x > 0 ? x : 300.0; }();
  return v;
}

void Engine::Barrier(const char *what) {
  if (!shared_ || cfg.num_worker <= 1) return;
  const uint32_t gen = shared_->generation.load(std::memory_order_acquire);
  const uint32_t here = shared_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1;
  if (here == (uint32_t)cfg.num_worker) { // last one in: re-arm, then release the others
    shared_->arrived.store(0, std::memory_order_relaxed);
    shared_->generation.fetch_add(1, std::memory_order_release);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spin = 0;; ++spin) {
    if (shared_->generation.load(std::memory_order_acquire) != gen) return;
    if (spin < 4096) continue; // a step barrier is usually released within microseconds
    if ((spin & 63) == 0) {
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (waited > ipc_timeout_s())
        fatal(__FILE__, __LINE__, "worker " + std::to_string(worker_id_) + " (device " + std::to_string(device_) + ") waited " +
                                      std::to_string((int)waited) + " s at the worker barrier '" + what + "': " +
                                      std::to_string(shared_->arrived.load()) + " of " + std::to_string(cfg.num_worker) +
                                      " workers arrived -- a worker died or is stuck (SAMGRAPH_IPC_TIMEOUT_S)");
    }
    spin < 65536 ? (void)sched_yield() : (void)usleep(200);
  }
}

// hipIpcOpenMemHandle under a deadline: the call runs on a helper thread; if it has not returned in time the process
// ends with a message (the thread cannot be cancelled, so there is nothing to retry in-process).
void *Engine::OpenPeer(const hipIpcMemHandle_t &handle, uint32_t peer, size_t bytes, const char *what) {
  struct State { std::mutex m; std::condition_variable cv; bool done = false; hipError_t err = hipSuccess; void *ptr = nullptr; };
  auto st = std::make_shared<State>();
  const int dev = device_;
  std::thread([st, handle, dev] {
    void *p = nullptr;
    hipError_t e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipIpcOpenMemHandle(&p, handle, hipIpcMemLazyEnablePeerAccess);
    std::lock_guard<std::mutex> g(st->m);
    st->err = e;
    st->ptr = p;
    st->done = true;
    st->cv.notify_all();
  }).detach();
  std::unique_lock<std::mutex> lk(st->m);
  if (!st->cv.wait_for(lk, std::chrono::duration<double>(ipc_timeout_s()), [&] { return st->done; }))
    fatal(__FILE__, __LINE__, std::string("hipIpcOpenMemHandle(") + what + ") did not return within " +
                                  std::to_string((int)ipc_timeout_s()) + " s: worker " + std::to_string(worker_id_) + " (device " +
                                  std::to_string(device_) + ") opening the shard of worker " + std::to_string(peer) + ", " +
                                  std::to_string(bytes) + " bytes (SAMGRAPH_IPC_TIMEOUT_S)");
  SAM_CHECK(st->err == hipSuccess, std::string("hipIpcOpenMemHandle(") + what + ") from worker " + std::to_string(peer) + ": " +
                                       hipGetErrorString(st->err));
  return st->ptr;
}

// ------------------------------------------------------------------ shuffler
// GPUShuffler (cuda/cuda_shuffler.cc:38-160) for one worker, DistAlignedShuffler
// (dist/dist_shuffler_aligned.cc:37-146) for arch6: same Fisher-Yates with
// std::default_random_engine(seed) + uniform_int_distribution<size_t>(i, n-1).
void Engine::ShufflerInit() {
  const uint32_t *train = (const uint32_t *)ds.train_set.ptr;
  const size_t nw = cfg.arch == kArch6 ? cfg.num_worker : 1;
  const size_t origin = ds.num_train;
  num_data_ = (origin + nw - 1) / nw * nw; // aligned to num_worker (:46)
  shuf_host_.assign(train, train + origin);
  for (size_t i = 0; i < num_data_ - origin; ++i) shuf_host_.push_back(train[i]); // :52-54
  num_local_data_ = num_data_ / nw;
  num_local_step_ = (num_local_data_ + cfg.batch_size - 1) / cfg.batch_size;
  num_global_step_ = num_local_step_ * nw;
  global_step_offset_ = num_local_step_ * worker_id_;
  global_data_offset_ = num_local_data_ * worker_id_;
  cur_epoch_ = 0;
  cur_step_ = num_local_step_;
  shuf_initialized_ = false;
  SAM_HIP(hipMalloc((void **)&shuf_dev_, std::max<size_t>(1, num_local_data_) * 4));
  { // a train SET: no node twice (what lets a batch promise distinct seeds to the sampler)
    std::vector<bool> seen(ds.num_node, false);
    train_distinct_ = true;
    for (size_t i = 0; i < origin && train_distinct_; ++i) {
      if (train[i] >= ds.num_node || seen[train[i]]) train_distinct_ = false;
      else seen[train[i]] = true;
    }
  }
}

// both copies of a padding node inside [offset, offset + size) of this worker's slice?
bool Engine::BatchSeedsDistinct(size_t offset, size_t size) const {
  if (!train_distinct_) return false;
  for (const auto &pr : pad_pairs_)
    if (pr.first >= offset && pr.first < offset + size && pr.second >= offset && pr.second < offset + size) return false;
  return true;
}

void Engine::Reshuffle() {
  if (!shuf_initialized_) { cur_epoch_ = 0; shuf_initialized_ = true; } else { cur_epoch_++; }
  cur_step_ = 0;
  if (cur_epoch_ >= cfg.num_epoch) return;
  uint64_t seed;
  if (cfg.arch == kArch6) seed = cur_epoch_;   // all samplers share the permutation (:92-94)
  else if (cfg.has_seed) seed = cfg.seed + cur_epoch_;
  else seed = std::chrono::system_clock::now().time_since_epoch().count(); // cuda_shuffler.cc:89
  auto g = std::default_random_engine(seed);
  uint32_t *data = shuf_host_.data();
  for (size_t i = 0; num_data_ && i < num_data_ - 1; i++) {
    std::uniform_int_distribution<size_t> d(i, num_data_ - 1);
    std::swap(data[i], data[d(g)]);
  }
  // where the padding copies went (at most num_worker - 1 nodes appear twice in the aligned epoch)
  pad_pairs_.clear();
  if (num_data_ > ds.num_train) {
    const uint32_t *train = (const uint32_t *)ds.train_set.ptr;
    std::unordered_map<uint32_t, size_t> first;
    for (size_t i = 0; i < num_data_ - ds.num_train; ++i) first[train[i]] = (size_t)-1;
    for (size_t pos = 0; pos < num_data_; ++pos) {
      auto it = first.find(data[pos]);
      if (it == first.end()) continue;
      if (it->second == (size_t)-1) { it->second = pos; continue; }
      const size_t a = it->second, b = pos, lo = global_data_offset_, hi = global_data_offset_ + num_local_data_;
      if (a >= lo && a < hi && b >= lo && b < hi) pad_pairs_.emplace_back(a - lo, b - lo);
    }
  }
  // the previous epoch's batches may still be copying their seeds out of shuf_dev_ on the pipeline streams
  for (auto &P : pipes_)
    if (P.stream) SAM_HIP(hipStreamSynchronize(P.stream));
  SAM_HIP(hipMemcpyAsync(shuf_dev_, data + global_data_offset_, num_local_data_ * 4, hipMemcpyHostToDevice, stream_));
  SAM_HIP(hipStreamSynchronize(stream_));
}

// SAMGRAPH_SANITY_CHECK (cuda_shuffler.cc:147-154): no invalid id in the batch (GPUSanityCheckList) and no train node
// handed out twice within an epoch (GPUBatchSanityCheck).  The shuffled train set has a host copy, so the check
// runs there; a violation is fatal, as the device-side asserts of the reference are.
void Engine::SanityCheckBatch(const uint32_t *seeds, size_t n) {
  if (cur_step_ == 0) sanity_seen_.assign(ds.num_node, false);
  for (size_t i = 0; i < n; ++i) {
    SAM_CHECK(seeds[i] != GGMS_EMPTY_KEY && seeds[i] < ds.num_node, "sanity check: invalid node id in a batch");
    SAM_CHECK(!sanity_seen_[seeds[i]], "sanity check: a train node was handed out twice in one epoch");
    sanity_seen_[seeds[i]] = true;
  }
}

bool Engine::ShufflerNext(Batch *b, hipStream_t copy_stream) {
  cur_step_++;
  if (cur_step_ >= num_local_step_) Reshuffle();
  if (cur_epoch_ >= cfg.num_epoch) return false;
  const size_t offset = cur_step_ * cfg.batch_size;
  SAM_CHECK(offset < num_local_data_, "shuffler offset out of range");
  size_t size = (offset + cfg.batch_size > num_local_data_) ? (num_local_data_ - offset) : cfg.batch_size;
  if (cfg.arch == kArch6 && cur_epoch_ == 0 && cur_step_ == 0) { // first batch x1.25, :137-140
    size = (size_t)(size * 1.25);
    size = (offset + size > num_local_data_) ? (num_local_data_ - offset) : size;
  }
  b->num_seeds = size;
  b->key = BatchKey(cur_epoch_, global_step_offset_ + cur_step_);
  static const bool sanity = getenv("SAMGRAPH_SANITY_CHECK") != nullptr; // run_config.cc:126-128
  if (sanity && cfg.arch == kArch1) SanityCheckBatch(shuf_host_.data() + global_data_offset_ + offset, size);
  SAM_HIP(hipMemcpyAsync(b->output_nodes, shuf_dev_ + offset, size * 4, hipMemcpyDeviceToDevice, copy_stream)); // Copy1D
  return true;
}

// ------------------------------------------------------------------ device graph (GGMS topology)
static void *dev_upload(const void *host, size_t bytes, hipStream_t s) {
  void *d = nullptr;
  // shards are published to the other workers with hipIpc: sized so that a peer can open them (include/ggms.h)
  SAM_HIP(hipMalloc(&d, ggms_ipc_safe_bytes(std::max<size_t>(bytes, 16))));
  if (bytes) SAM_HIP(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, s));
  return d;
}

static const void *map_host(void *host, size_t bytes) {
  // cudaHostRegister(..., ReadOnly) + zero-copy reads (dist_engine.cc:217-241)
  if (bytes == 0) return nullptr;
  SAM_HIP(hipHostRegister(host, bytes, hipHostRegisterMapped));
  void *d = nullptr;
  SAM_HIP(hipHostGetDevicePointer(&d, host, 0));
  return d;
}

void Engine::UploadGraph() {
  const uint32_t *indptr = (const uint32_t *)ds.indptr.ptr, *indices = (const uint32_t *)ds.indices.ptr;
  std::memset(&graph_, 0, sizeof(graph_));
  graph_.num_node = (uint32_t)ds.num_node;
  if (!cfg.use_dist_graph) { // dist_engine.cc:203-216 / cuda_engine: whole CSR on this GPU
    d_indptr_ = (uint32_t *)dev_upload(indptr, ds.indptr.bytes, stream_);
    d_indices_ = (uint32_t *)dev_upload(indices, ds.indices.bytes, stream_);
    graph_.indptr = d_indptr_;
    graph_.indices = d_indices_;
    SAM_HIP(hipStreamSynchronize(stream_));
    return;
  }
  // DistGraph::GraphLoad, cuda/dist_graph.cu:309-385
  const uint32_t P = (uint32_t)cfg.num_worker, p = (uint32_t)worker_id_;
  const uint32_t num_cache_edge = (uint32_t)(ds.num_edge * cfg.dist_graph_percentage); // dist_engine.cc:225
  uint32_t num_cache_node = 0;
  while (num_cache_node < ds.num_node && indptr[num_cache_node] < num_cache_edge) ++num_cache_node;
  // _DatasetPartition :228-272: shard p = nodes v == p (mod P), v < num_cache_node
  const size_t isz = num_cache_node / P + (p < num_cache_node % P ? 1 : 0) + 1;
  std::vector<uint32_t> pip(isz);
  size_t ecount = 0;
  for (uint32_t v = p; v < num_cache_node; v += P) ecount += indptr[v + 1] - indptr[v];
  std::vector<uint32_t> pix(std::max<size_t>(ecount, 1));
  uint32_t cnt = 0;
  for (uint32_t v = p; v < num_cache_node; v += P) {
    const uint32_t ne = indptr[v + 1] - indptr[v];
    pip[v / P] = cnt;
    std::memcpy(&pix[cnt], &indices[indptr[v]], ne * 4ull);
    cnt += ne;
  }
  pip[isz - 1] = cnt;
  part_indptr_.assign(P + 1, nullptr);
  part_indices_.assign(P + 1, nullptr);
  part_indptr_[p] = dev_upload(pip.data(), isz * 4, stream_);
  part_indices_[p] = dev_upload(pix.data(), ecount * 4, stream_);
  SAM_HIP(hipStreamSynchronize(stream_));
  // _DataIpcShare :274-307: publish, barrier, open peers, barrier
  if (P > 1) {
    SAM_HIP(hipIpcGetMemHandle(&shared_->graph_indptr[p], part_indptr_[p]));
    SAM_HIP(hipIpcGetMemHandle(&shared_->graph_indices[p], part_indices_[p]));
    shared_->indptr_words[p] = isz;
    shared_->indices_words[p] = ecount;
    Barrier("graph shards published");
    for (uint32_t q = 0; q < P; ++q) {
      if (q == p) continue;
      part_indptr_[q] = OpenPeer(shared_->graph_indptr[q], q, shared_->indptr_words[q] * 4, "graph indptr shard");
      part_indices_[q] = OpenPeer(shared_->graph_indices[q], q, shared_->indices_words[q] * 4, "graph indices shard");
    }
    Barrier("graph shards opened");
  }
  // slot P: the whole CSR, :367-381.  Its neighbour lists stay in (device-mapped) host memory as in the reference; its
  // `indptr` -- 4 B per node, 0.44 GB at papers100M size against 288 GB of HBM -- is kept on the GPU: a seed beyond
  // num_cache_node then pays ONE PCIe round trip (its sampled positions) instead of two dependent ones (list head, then
  // positions), and 40 % fewer PCIe reads per batch.  The layout of slot P is not observable through the interface.
  part_indptr_[P] = dev_upload(ds.indptr.ptr, ds.indptr.bytes, stream_);
  SAM_HIP(hipStreamSynchronize(stream_));
  part_indices_[P] = (void *)map_host(ds.indices.ptr, ds.indices.bytes);
  // the P + 1 pointers stay on the host: ggms_sample_batch hands them to its kernels by value (include/ggms.h)
  SAM_CHECK(P <= GGMS_MAX_PARTS, "use_dist_graph: at most GGMS_MAX_PARTS topology shards");
  graph_.part_indptr = (const ggms_id_t *const *)part_indptr_.data();
  graph_.part_indices = (const ggms_id_t *const *)part_indices_.data();
  graph_.num_part = P;
  graph_.num_cache_node = num_cache_node;
}

// ------------------------------------------------------------------ init
void Engine::SampleInit(int worker_id, const std::string &ctx) {
  SAM_CHECK(data_ready_, "samgraph_data_init first");
  worker_id_ = worker_id;
  device_ = parse_device(ctx);
  SAM_HIP(hipSetDevice(device_));
  SAM_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  SAM_HIP(hipStreamCreateWithFlags(&stream_extract_, hipStreamNonBlocking));
  if (cfg.extract_streams > 1) SAM_HIP(hipStreamCreateWithFlags(&stream_extract2_, hipStreamNonBlocking));
  UploadGraph();
  ShufflerInit();
  const uint32_t L = (uint32_t)cfg.fanout.size();
  // first batch of arch6 is x1.25 (dist_shuffler_aligned.cc:137-140): size every buffer for it
  max_seeds_ = (size_t)(cfg.batch_size * 1.25) + 1;
  max_input_.resize(L);
  max_edges_.resize(L);
  SAM_GGMS(ggms_sample_batch_capacity(max_seeds_, cfg.fanout.data(), L, max_input_.data(), max_edges_.data(), &max_unique_));
  // OrderedHashTable(PredictNumNodes(...)) dist_engine.cc:423-424; direct layout by default
  std::memset(&ht_, 0, sizeof(ht_));
  ht_.direct = cfg.direct_table ? 1 : 0;
  ht_.o2n_size = cfg.direct_table ? ds.num_node : ggms_hashtable_num_buckets(max_unique_);
  ht_.n2o_size = max_unique_;
  SAM_HIP(hipMalloc(&ht_.o2n, ht_.o2n_size * (cfg.direct_table ? 8 : 16)));
  SAM_HIP(hipMalloc((void **)&ht_.n2o, max_unique_ * 4));
  SAM_HIP(hipMalloc((void **)&ht_.num_items_dev, 16));
  SAM_GGMS(ggms_hashtable_init(&ht_, stream_));
  std::memset(&extra_, 0, sizeof(extra_));
  if (cfg.sample_type == GGMS_WEIGHTED_KHOP || cfg.sample_type == GGMS_WEIGHTED_KHOP_HASH_DEDUP) { // dist_engine.cc:210-213
    d_prob_ = dev_upload(ds.prob_table.ptr, ds.prob_table.bytes, stream_);
    d_alias_ = dev_upload(ds.alias_table.ptr, ds.alias_table.bytes, stream_);
    extra_.prob_table = (const float *)d_prob_;
    extra_.alias_table = (const ggms_id_t *)d_alias_;
  }
  if (cfg.sample_type == GGMS_WEIGHTED_KHOP_PREFIX) {
    d_prob_ = dev_upload(ds.prob_table.ptr, ds.prob_table.bytes, stream_);
    extra_.prob_table = (const float *)d_prob_;
  }
  extra_.random_walk_length = cfg.random_walk_length;
  extra_.random_walk_restart_prob = cfg.random_walk_restart_prob;
  extra_.num_random_walk = cfg.num_random_walk;
  // GPURandomStates dist_engine.cc:432-433; seed = wall clock unless the "seed" key is given
  num_states_ = ggms_random_states_count(cfg.sample_type, cfg.fanout.data(), L, max_seeds_, cfg.num_random_walk);
  size_t max_in = 0;
  for (auto v : max_input_) max_in = std::max(max_in, v);
  num_states_ = std::max(num_states_, (max_in + 127) / 128 * 8);
  num_states_ = std::max(num_states_, (max_in + 1023) / 1024 * 256); // khop2: one stream per thread of a 1024-seed tile
  if (cfg.sample_type == GGMS_RANDOM_WALK)
    num_states_ = std::max(num_states_, ggms_random_walk_num_states(max_in, cfg.num_random_walk));
  SAM_HIP(hipMalloc(&states_, num_states_ * GGMS_RNG_STATE_BYTES));
  const uint64_t seed = cfg.has_seed ? cfg.seed + 1000003ull * worker_id
                                     : (uint64_t)std::chrono::system_clock::now().time_since_epoch().count();
  SAM_GGMS(ggms_random_states_init(states_, num_states_, seed, stream_));
  ws_bytes_ = ggms_sample_batch_workspace_bytes(cfg.sample_type, max_seeds_, cfg.fanout.data(), L, &extra_);
  SAM_HIP(hipMalloc(&ws_, ws_bytes_));
  // pipeline 0 = {stream_, ht_, ws_}; the others get their own stream, table and workspace
  pipes_.assign(cfg.pipelines, Pipe{});
  for (size_t p = 0; p < pipes_.size(); ++p) {
    Pipe &P = pipes_[p];
    SAM_HIP(hipEventCreateWithFlags(&P.rng_done, hipEventDisableTiming));
    if (p == 0) {
      P.stream = stream_;
      P.ht = ht_;
      P.ws = ws_;
      continue;
    }
    SAM_HIP(hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
    P.ht = ht_;
    SAM_HIP(hipMalloc(&P.ht.o2n, ht_.o2n_size * (cfg.direct_table ? 8 : 16)));
    SAM_HIP(hipMalloc((void **)&P.ht.n2o, max_unique_ * 4));
    SAM_HIP(hipMalloc((void **)&P.ht.num_items_dev, 16));
    SAM_GGMS(ggms_hashtable_init(&P.ht, stream_));
    SAM_HIP(hipMalloc(&P.ws, ws_bytes_));
  }
  SAM_HIP(hipStreamSynchronize(stream_));
  prof.Resize(cfg.num_epoch, num_global_step_);
  if (cfg.UsePresample()) { // dist_engine.cc:455-466: worker 0 ranks the nodes, everybody waits
    auto t0 = std::chrono::steady_clock::now();
    if (worker_id_ == 0) Presample();
    Barrier("presample ranking");
    prof.LogInit(/*kLogInitL2Presample*/ 8, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  sample_ready_ = true;
}

// PreSampler (dist/pre_sampler.cc:39-139): sample `presample_epoch` epochs of the WHOLE train set with a
// GPUShuffler of its own, count how often each node is an input node, rank by (freq << 32 | id) descending.
// Counting happens on the device (one atomicAdd per input node) instead of D2H copy + OpenMP loop.
void Engine::Presample() {
  const uint32_t L = (uint32_t)cfg.fanout.size();
  const size_t n_train = ds.num_train;
  const size_t steps = (n_train + cfg.batch_size - 1) / cfg.batch_size; // drop_last = false, :41-42
  std::vector<uint32_t> data((const uint32_t *)ds.train_set.ptr, (const uint32_t *)ds.train_set.ptr + n_train);
  uint32_t *d_train = nullptr, *d_freq = nullptr;
  uint64_t *d_counts = nullptr;
  SAM_HIP(hipMalloc((void **)&d_train, std::max<size_t>(n_train, 1) * 4));
  SAM_HIP(hipMalloc((void **)&d_freq, ds.num_node * 4));
  SAM_HIP(hipMalloc((void **)&d_counts, (3 * L + 8) * 8));
  SAM_HIP(hipMemsetAsync(d_freq, 0, ds.num_node * 4, stream_));
  std::vector<uint32_t *> row(L), col(L), dat(L, nullptr);
  for (uint32_t i = 0; i < L; ++i) {
    SAM_HIP(hipMalloc((void **)&row[i], std::max<size_t>(max_edges_[i], 4) * 4));
    SAM_HIP(hipMalloc((void **)&col[i], std::max<size_t>(max_edges_[i], 4) * 4));
    if (cfg.sample_type == GGMS_RANDOM_WALK) SAM_HIP(hipMalloc((void **)&dat[i], std::max<size_t>(max_edges_[i], 4) * 4));
  }
  for (size_t e = 0; e < cfg.presample_epoch; ++e) {
    const uint64_t seed = cfg.has_seed ? cfg.seed + 0x5a5a5aull + e
                                       : (uint64_t)std::chrono::system_clock::now().time_since_epoch().count();
    auto g = std::default_random_engine(seed); // GPUShuffler::ReShuffle, cuda_shuffler.cc:89-110
    for (size_t i = 0; n_train && i < n_train - 1; i++) {
      std::uniform_int_distribution<size_t> d(i, n_train - 1);
      std::swap(data[i], data[d(g)]);
    }
    SAM_HIP(hipMemcpyAsync(d_train, data.data(), n_train * 4, hipMemcpyHostToDevice, stream_));
    for (size_t s = 0; s < steps; ++s) {
      const size_t off = s * cfg.batch_size, size = std::min(cfg.batch_size, n_train - off);
      ggms_sample_extra_t extra = extra_;
      extra.data = dat.data();
      extra.seeds_distinct = train_distinct_ ? 1u : 0u; // slices of a permutation of the train set
      // pipeline 0's table: its version stamp keeps counting when the training batches follow
      SAM_GGMS(ggms_sample_batch(cfg.sample_type, &graph_, d_train + off, size, cfg.fanout.data(), L, &pipes_[0].ht,
                                 states_, num_states_, row.data(), col.data(), d_counts, &extra, ws_, ws_bytes_, stream_));
      SAM_GGMS(ggms_count_nodes(d_freq, pipes_[0].ht.n2o, max_unique_, d_counts + 3 * L, stream_));
    }
    SAM_HIP(hipStreamSynchronize(stream_)); // `data` is reshuffled on the host next
  }
  std::vector<uint32_t> freq(ds.num_node);
  SAM_HIP(hipMemcpy(freq.data(), d_freq, ds.num_node * 4, hipMemcpyDeviceToHost));
  std::vector<uint64_t> keys(ds.num_node);
  for (size_t i = 0; i < ds.num_node; ++i) keys[i] = ((uint64_t)freq[i] << 32) | (uint64_t)i; // :45-50
  std::sort(keys.begin(), keys.end(), std::greater<uint64_t>());                            // :113-119
  uint32_t *rank = (uint32_t *)ds.ranking_nodes.ptr;
  for (size_t i = 0; i < ds.num_node; ++i) rank[i] = (uint32_t)keys[i]; // GetRankNode :141-151
  for (uint32_t i = 0; i < L; ++i) { (void)hipFree(row[i]); (void)hipFree(col[i]); if (dat[i]) (void)hipFree(dat[i]); }
  (void)hipFree(d_train); (void)hipFree(d_freq); (void)hipFree(d_counts);
}

void Engine::BuildCache() {
  const size_t row_bytes = ds.feat_dim * ggms_dtype_bytes(ds.feat_dtype);
  const char *feat = (const char *)ds.feat.ptr;
  // labels are 8 B x N: always resident on the device (the reference keeps them on the host and
  // gathers through zero-copy in gpu_extract mode, dist_loops.cc:938-974)
  label_src_ = dev_upload(ds.label.ptr, ds.label.bytes, stream_);
  if (!cfg.UseGPUCache()) {
    if (cfg.arch == kArch6 && !cfg.gpu_extract) {
      // cache 0 without gpu_extract: the table stays in host memory and every row takes the host-staged path
      // (DoIdCopy + DoCPUFeatureExtract + DoFeatureCopy, dist_loops_arch6.cc:111-133; StagedExtract)
      feat_src_ = nullptr;
    } else if (cfg.arch == kArch1) {
      // arch1: the whole table lives in HBM and is gathered directly (cuda_loops_arch1.cc:61)
      d_feat_ = dev_upload(feat, ds.feat.bytes, stream_);
      feat_src_ = d_feat_;
    } else {
      feat_src_ = map_host(ds.feat.ptr, ds.feat.bytes); // cache 0 %: DoGPUFeatureExtract from host, dist_loops.cc:585-634
    }
    SAM_HIP(hipStreamSynchronize(stream_));
    return;
  }
  // GPUCacheManager ctor: cuda_cache_manager_host.cc:61-130 (replicated) / :133-254 (partition)
  const uint32_t *rank_in = (const uint32_t *)ds.ranking_nodes.ptr;
  num_cached_nodes_ = (size_t)(ds.num_node * cfg.cache_percentage);
  std::vector<uint32_t> rank(rank_in, rank_in + ds.num_node);
  const uint32_t P = cfg.part_cache ? (uint32_t)cfg.num_worker : 1, p = cfg.part_cache ? (uint32_t)worker_id_ : 0;
  // hybrid store: slots [0, R) = the hottest cached nodes, a copy on every GPU; slots [R, num_cached) sharded
  num_replica_ = cfg.part_cache ? (size_t)(num_cached_nodes_ * cfg.replicate_percentage) : 0;
  SAM_CHECK(num_replica_ == 0 || cfg.gpu_extract, "replicate_percentage needs gpu_extract (one fused gather over all tiers)");
  const size_t R = num_replica_;
  if (cfg.part_cache) { // load balance across shards: :169-171 (the replicated prefix keeps its rank order)
    std::mt19937 eg((uint32_t)num_cached_nodes_);
    std::shuffle(rank.begin() + R, rank.begin() + num_cached_nodes_, eg);
  }
  // Everything cached (cache_percentage 1.0; 288 GB of HBM hold every BASELINE feature table): rows stay in NODE
  // order, slot = node id -- no id -> slot table and no table read per gathered row (ggms_extract_cached with
  // table == NULL).  The reference ranks and shuffles even then; the layout is not observable through its interface.
  const bool full_cache = num_cached_nodes_ == ds.num_node && R == 0;
  if (full_cache) {
    for (size_t i = 0; i < ds.num_node; ++i) rank[i] = (uint32_t)i;
    cache_table_ = nullptr;
  } else {
    std::vector<uint32_t> table(ds.num_node, GGMS_EMPTY_KEY);
    for (size_t i = 0; i < num_cached_nodes_; ++i) table[rank[i]] = (uint32_t)i; // :197-229
    cache_table_ = (uint32_t *)dev_upload(table.data(), ds.num_node * 4, stream_);
  }
  // Rows are staged through a bounded host buffer (64 MB at a time), never through a host copy of the whole replica
  // or shard: at papers100M size a replica is 46 GB, and eight workers of one node would hold eight of them at once.
  // ... gathered by the host team (omp_thread_num threads) into TWO pinned buffers: the cores fill one while the copy
  // engine drains the other (one thread and one pageable buffer took a minute per 46-GB replica).
  Team team((int)std::max<size_t>(1, cfg.omp_thread_num));
  const size_t step = std::max<size_t>(1, (64u << 20) / row_bytes);
  char *stage[2] = {nullptr, nullptr};
  hipEvent_t drained[2];
  for (int k = 0; k < 2; ++k) {
    SAM_HIP(hipHostMalloc((void **)&stage[k], step * row_bytes + 16));
    SAM_HIP(hipEventCreateWithFlags(&drained[k], hipEventDisableTiming));
  }
  auto upload_rows = [&](size_t first, size_t stride, size_t count) -> void * {
    void *d = nullptr;
    // shards are published to the other workers with hipIpc: sized so that a peer can open them (include/ggms.h)
    SAM_HIP(hipMalloc(&d, ggms_ipc_safe_bytes(std::max<size_t>(count * row_bytes, 16))));
    if (ds.feat_is_zero) {
      // rows of a table that is known to be all zero need no gather: touching them in rank order would fault in every
      // page of the untouched anonymous mapping once per worker (two minutes of page faults at papers100M size)
      SAM_HIP(hipMemsetAsync(d, 0, count * row_bytes, stream_));
      SAM_HIP(hipStreamSynchronize(stream_));
      return d;
    }
    size_t k = 0;
    for (size_t lo = 0; lo < count; lo += step, ++k) {
      const size_t m = std::min(step, count - lo);
      char *buf = stage[k & 1];
      if (k >= 2) SAM_HIP(hipEventSynchronize(drained[k & 1])); // the copy that last read this buffer is done
      team.ParallelFor(m, [&](size_t a, size_t b, int) {
        for (size_t i = a; i < b; ++i)
          std::memcpy(buf + i * row_bytes, feat + (size_t)(rank[first + (lo + i) * stride] & ds.feat_mask) * row_bytes, row_bytes);
      });
      SAM_HIP(hipMemcpyAsync((char *)d + lo * row_bytes, buf, m * row_bytes, hipMemcpyHostToDevice, stream_));
      SAM_HIP(hipEventRecord(drained[k & 1], stream_));
    }
    SAM_HIP(hipStreamSynchronize(stream_));
    return d;
  };
  if (R) d_replica_ = upload_rows(0, 1, R); // this GPU's copy of the hottest rows
  // DistGraph::FeatureLoad / _PartitionFeature, dist_graph.cu:493-521: rows rank[i], (i - R) == p (mod P)
  const size_t sharded = num_cached_nodes_ - R;
  const size_t my_rows = sharded / P + (p < sharded % P ? 1 : 0);
  cache_parts_.assign(P, nullptr);
  cache_parts_[p] = upload_rows(R + p, P, my_rows);
  SAM_HIP(hipStreamSynchronize(stream_));
  if (P > 1) { // _DataIpcShare
    SAM_HIP(hipIpcGetMemHandle(&shared_->feat_part[p], cache_parts_[p]));
    shared_->feat_rows[p] = my_rows;
    Barrier("feature shards published");
    for (uint32_t q = 0; q < P; ++q)
      if (q != p) cache_parts_[q] = OpenPeer(shared_->feat_part[q], q, shared_->feat_rows[q] * row_bytes, "feature shard");
    Barrier("feature shards opened");
  }
  SAM_CHECK(P <= GGMS_MAX_PARTS, "part_cache: at most GGMS_MAX_PARTS feature shards");
  num_cache_part_ = cfg.part_cache ? P : 0;
  for (int k = 0; k < 2; ++k) {
    (void)hipHostFree(stage[k]);
    (void)hipEventDestroy(drained[k]);
  }
  // miss tier: pinned host memory read by the gather kernel itself (GPUExtractMissData, :573-625); the host-staged
  // path (`gpu_extract` off) reads the table with the host cores instead and needs no device mapping of it
  feat_src_ = cfg.gpu_extract ? map_host(ds.feat.ptr, ds.feat.bytes) : nullptr;
  SAM_HIP(hipStreamSynchronize(stream_));
}

void Engine::TrainInit(int worker_id, const std::string &ctx) {
  SAM_CHECK(sample_ready_, "samgraph_sample_init first");
  SAM_CHECK(parse_device(ctx) == device_, "arch6: sampler and trainer share the GPU (cuda_cache_manager_host.cc:152-155)");
  (void)worker_id;
  SAM_HIP(hipSetDevice(device_));
  auto t0 = std::chrono::steady_clock::now();
  BuildCache();
  prof.LogInit(/*kLogInitL2BuildCache*/ 10, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  if (getenv("SAMGRAPH_LOG_NODE_ACCESS") || getenv("SAMGRAPH_LOG_NODE_ACCESS_SIMPLE")) { // run_config.cc:118-124
    SAM_HIP(hipMalloc((void **)&node_access_dev_, ds.num_node * 4));
    SAM_HIP(hipMemset(node_access_dev_, 0, ds.num_node * 4));
  }
  // batch slots (GraphPool(max_copying_jobs), cuda_engine.cc:151): buffers sized once at their bounds
  const uint32_t L = (uint32_t)cfg.fanout.size();
  const size_t row_bytes = ds.feat_dim * ggms_dtype_bytes(ds.feat_dtype);
  size_t nslots = 2;
  if (cfg.raw.count("max_copying_jobs")) nslots = std::max<size_t>(2, std::min<size_t>(4, std::stoull(cfg.raw["max_copying_jobs"]) + 1));
  nslots = std::max(nslots, 2 + cfg.lookahead); // the trainer's batch + the one asked for + the ones enqueued ahead
  for (size_t s = 0; s < nslots; ++s) {
    auto b = std::make_unique<Batch>();
    b->slot = (int)s;
    b->row.resize(L); b->col.resize(L); b->data.resize(L, nullptr);
    for (uint32_t i = 0; i < L; ++i) {
      SAM_HIP(hipMalloc((void **)&b->row[i], std::max<size_t>(max_edges_[i], 4) * 4));
      SAM_HIP(hipMalloc((void **)&b->col[i], std::max<size_t>(max_edges_[i], 4) * 4));
      if (cfg.sample_type == GGMS_RANDOM_WALK) SAM_HIP(hipMalloc((void **)&b->data[i], std::max<size_t>(max_edges_[i], 4) * 4));
    }
    SAM_HIP(hipMalloc((void **)&b->input_nodes, max_unique_ * 4));
    SAM_HIP(hipMalloc((void **)&b->output_nodes, max_seeds_ * 4));
    SAM_HIP(hipMalloc(&b->feat, max_unique_ * row_bytes));
    SAM_HIP(hipMalloc((void **)&b->label, max_seeds_ * 8));
    SAM_HIP(hipMalloc((void **)&b->counts_dev, (3 * L + 8) * 8));
    SAM_HIP(hipMemset(b->counts_dev, 0, (3 * L + 8) * 8));
    if (StagedHostTier()) { // index arrays of GetMissCacheIndex + pinned / device staging of the miss rows
      if (cache_table_) { // (no cache: no split, and the rows land in the batch's feature buffer directly)
        for (uint32_t **p : {&b->miss_src, &b->miss_dst, &b->hit_src, &b->hit_dst}) SAM_HIP(hipMalloc((void **)p, max_unique_ * 4));
        SAM_HIP(hipMalloc(&b->idx_ws, ggms_cache_index_workspace_bytes(max_unique_)));
        SAM_HIP(hipMalloc(&b->miss_rows_dev, max_unique_ * row_bytes));
      }
      SAM_HIP(hipEventCreateWithFlags(&b->ev_ids, hipEventDisableTiming));
      SAM_HIP(hipHostMalloc(&b->miss_rows_host, max_unique_ * row_bytes));
      SAM_HIP(hipHostMalloc((void **)&b->miss_ids_host, max_unique_ * 4));
    }
    SAM_HIP(hipHostMalloc((void **)&b->counts, (3 * L + 8) * 8));
    std::memset(b->counts, 0, (3 * L + 8) * 8);
    SAM_HIP(hipEventCreateWithFlags(&b->ev_seeds, hipEventDisableTiming));
    SAM_HIP(hipEventCreateWithFlags(&b->ev_label, hipEventDisableTiming));
    SAM_HIP(hipEventCreate(&b->ev_start));
    SAM_HIP(hipEventCreate(&b->ev_sampled));
    SAM_HIP(hipEventCreate(&b->ev_xstart));
    SAM_HIP(hipEventCreate(&b->ev_done));
    SAM_GGMS(ggms_launch_timer_create(&b->gather_timer));
    slots_.push_back(std::move(b));
  }
  train_ready_ = true;
}

void Engine::Init() { // samgraph_init, single process
  if (cfg.arch == kArch0) {
    CpuInit();
    return;
  }
  DataInit();
  const std::string ctx = "cuda:" + std::to_string(cfg.sampler_device);
  SampleInit(0, ctx);
  TrainInit(0, "cuda:" + std::to_string(cfg.trainer_device));
}

void Engine::Start() {}

void Engine::Shutdown() {
  shutdown_ = true;
  bg_stop_ = true;
  pool_cv_.notify_all();
  if (bg_.joinable()) bg_.join();
  if (cfg.arch == kArch0) CpuShutdown();
  for (auto &P : pipes_)
    if (P.stream) (void)hipStreamSynchronize(P.stream);
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (stream_extract_) (void)hipStreamSynchronize(stream_extract_);
  if (stream_extract2_) (void)hipStreamSynchronize(stream_extract2_);
}

// ------------------------------------------------------------------ hot loop
Batch *Engine::AcquireSlot(bool background) {
  for (;;) {
    {
      std::lock_guard<std::mutex> lk(pool_mu_);
      for (auto &b : slots_)
        if (!b->in_use && b->refs.load() == 0) { b->in_use = true; return b.get(); }
    }
    if (bg_stop_) return nullptr;
    // GraphPool full (cuda_loops_arch1.cc:45-48): the foreground call returns without sampling,
    // the background thread backs off and retries
    if (!background) return nullptr;
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

// samgraph_sample_once: the batches come out in shuffler order whoever asks, so the foreground call keeps
// `lookahead` more of them enqueued than it was asked for -- batch k+1 samples while batch k's rows are gathered and
// while the caller trains on batch k.  The pool hands them out in the same order (GetNextBatch).
void Engine::RunSampleOnce(bool background) {
  // the role is an argument, never inferred from the std::thread member (the new thread would read it while
  // ExtractStart is still assigning it)
  if (background) { // background loop: one per iteration
    (void)EnqueueOne(true);
    return;
  }
  ++fg_calls_;
  while (fg_enqueued_ < fg_calls_ + cfg.lookahead && EnqueueOne(false)) ++fg_enqueued_;
  // a call that found the pool full (cuda_loops_arch1.cc:45-48) stays owed: a later call catches up
}

// RunArch1LoopsOnce (cuda/cuda_loops_arch1.cc:43-86) / RunArch6LoopsOnce (dist/dist_loops_arch6.cc:236-243):
// shuffle -> sample -> extract, all enqueued with no host round trip.  false: no free slot, or training finished.
bool Engine::EnqueueOne(bool background) {
  if (cfg.arch == kArch0) return CpuEnqueueOne(background);
  SAM_CHECK(train_ready_, "engine not initialised");
  SAM_HIP(hipSetDevice(device_));
  Batch *b = AcquireSlot(background);
  if (!b) return false;
  // Consecutive batches go to the sampling pipelines round-robin; a batch's seeds are copied on ITS pipeline's
  // stream (not behind another pipeline's queued sampling).
  Pipe &P = pipes_[enq_count_ % pipes_.size()];
  hipStream_t ss = P.stream;
  if (!ShufflerNext(b, ss)) { // training finished
    std::lock_guard<std::mutex> lk(pool_mu_);
    b->in_use = false;
    return false;
  }
  const uint32_t L = (uint32_t)cfg.fanout.size();
  // The RNG pool -- and khop2's CSR -- is handed from batch to batch through rng_wait / rng_done, so the results
  // are those of the one-batch-at-a-time loop.
  ++enq_count_;
  SAM_HIP(hipEventRecord(b->ev_start, ss));
  ggms_sample_extra_t extra = extra_;
  extra.data = b->data.data();
  extra.seeds_distinct = BatchSeedsDistinct(cur_step_ * cfg.batch_size, b->num_seeds) ? 1u : 0u;
  if (pipes_.size() > 1 && cfg.sample_type != GGMS_KHOP0) {
    extra.rng_wait = last_rng_done_;
    extra.rng_done = P.rng_done;
    last_rng_done_ = P.rng_done;
  }
  // input nodes = the table's unique list (task->input_nodes, dist_loops.cc:357).  The table struct is plain data and
  // its n2o buffer the caller's: the batch builds the list directly in its slot (a later batch uses another slot)
  ggms_hashtable_t ht = P.ht;
  ht.n2o = b->input_nodes;
  ht.n2o_size = max_unique_;
  SAM_GGMS(ggms_sample_batch(cfg.sample_type, &graph_, b->output_nodes, b->num_seeds, cfg.fanout.data(), L, &ht,
                             states_, num_states_, b->row.data(), b->col.data(), b->counts_dev, &extra, P.ws, ws_bytes_,
                             ss));
  P.ht.version = ht.version; // the batch bumped the table's version stamp
  uint64_t *n_in = b->counts_dev + 3 * L, *n_miss = b->counts_dev + 3 * L + 2; // [3L + 1] = the batch's status word
  SAM_HIP(hipMemsetAsync(n_miss, 0, 8, ss));
  SAM_HIP(hipEventRecord(b->ev_sampled, ss));
  // DoGPULabelExtract (dist_loops.cc:938-974) needs the seeds only: it rides behind the batch on its sampling stream, not
  // between two gathers on the extract stream, which bounds the step.  (Not on a stream of its own: HIP streams share 4
  // hardware queues, and a fifth stream serialises streams that have nothing to do with each other.)
  SAM_GGMS(ggms_extract(b->label, label_src_, b->output_nodes, b->num_seeds, 1, GGMS_I64, ss));
  const bool mock = ds.feat_mask != 0xffffffffu; // SAMGRAPH_EMPTY_FEAT: host rows are node & mask
  // The extract stream bounds the step, and every event record / wait / small copy on it is a packet the command
  // processor works through between two gathers -- 28 us of dead time per 0.7-ms step with six of them
  // (profiles/r05_ab_extract_stream.txt).  LEAN batch: the gather writes nothing the host reads (no miss / tier
  // counters, no visit counts), so the batch's counts go to the host behind the label gather on the SAMPLING stream,
  // which has the slack, and the extract stream carries one wait and the gather -- whose start / end timestamps and
  // "rows are out" event ride on its own dispatch packet (b->gather_timer).  Otherwise the sequence the counters need.
  // (no row can miss when every node is cached: the miss count stays the zero the sampling stream wrote, and the
  // per-tier row counts, which nothing on the host reads, are not taken)
  const bool can_miss = num_cached_nodes_ < ds.num_node;
  const bool gather_counts = cfg.UseGPUCache() && (mock || ((num_replica_ || cache_table_) && can_miss));
  static const bool lean_off = [] { const char *e = getenv("SAMGRAPH_LEAN_EXTRACT"); return e && e[0] == '0'; }(); // A/B hook
  b->lean = !lean_off && !StagedHostTier() && !gather_counts && !node_access_dev_;
  if (b->lean) {
    SAM_HIP(hipMemcpyAsync(b->counts, b->counts_dev, (3 * L + 8) * 8, hipMemcpyDeviceToHost, ss));
    SAM_HIP(hipEventRecord(b->ev_done, ss)); // labels and counts are out; the rows: gather_timer
  } else {
    SAM_HIP(hipEventRecord(b->ev_label, ss));
  }
  // The gather is HBM-bound, the sampler latency-bound: they run on separate streams so that batch k's
  // extract overlaps batch k+1's sampling (the reference serialises them, dist_loops_arch6.cc:248-251)
  hipStream_t xs = (b->lean && stream_extract2_ && (enq_count_ & 1)) ? stream_extract2_ : stream_extract_;
  SAM_HIP(hipStreamWaitEvent(xs, b->ev_sampled, 0));
  if (b->lean) SAM_GGMS(ggms_launch_timer_arm(b->gather_timer));
  else SAM_HIP(hipEventRecord(b->ev_xstart, xs)); // the extract's own start: behind the previous batch's extract on xs
  if (StagedHostTier()) {
    StagedExtract(b, ss, xs);
  } else if (cfg.UseGPUCache() && (mock || num_replica_)) { // every tier in one gather; rows per tier counted
    if (gather_counts) SAM_HIP(hipMemsetAsync(n_miss, 0, 4 * 8, xs)); // {host, remote shard, local shard, replica} = counts[3L+2 .. 3L+5]
    ggms_feature_tiers_t tiers{};
    tiers.table = cache_table_;
    tiers.replica = d_replica_;
    tiers.num_replica = num_replica_;
    tiers.parts = (const void *const *)cache_parts_.data();
    tiers.num_part = std::max<uint32_t>(1, num_cache_part_);
    tiers.my_part = cfg.part_cache ? (uint32_t)worker_id_ : 0;
    tiers.host_feat = feat_src_;
    tiers.host_row_mask = mock ? ds.feat_mask : 0;
    SAM_GGMS(ggms_extract_tiered(b->feat, b->input_nodes, max_unique_, n_in, &tiers, ds.feat_dim, ds.feat_dtype,
                                 gather_counts ? n_miss : nullptr, xs));
  } else if (cfg.UseGPUCache()) {
    // DoArch6GetCacheMissIndex + DoArch6GPUCacheFeatureCopy (dist_loops.cc:1015-1285) in one pass; everything cached in
    // node order (no table): no row can miss, the count stays the zero the sampling stream wrote
    SAM_GGMS(ggms_extract_cached(b->feat, b->input_nodes, max_unique_, n_in, cache_table_,
                                 (const void *const *)cache_parts_.data(), num_cache_part_, feat_src_, ds.feat_dim,
                                 ds.feat_dtype, (cache_table_ && can_miss) ? n_miss : nullptr, xs));
  } else if (mock) { // GPUMockExtract, cuda_loops.cc:692-700 / dist_loops.cc:608-616
    SAM_GGMS(ggms_gather_scatter_masked(b->feat, feat_src_, b->input_nodes, nullptr, max_unique_, n_in, ds.feat_dim,
                                        ds.feat_dtype, ds.feat_mask, xs));
  } else {
    // DoGPUFeatureExtract (cuda/cuda_loops.cc, dist_loops.cc:585-634)
    SAM_GGMS(ggms_gather_scatter(b->feat, feat_src_, b->input_nodes, nullptr, max_unique_, n_in, ds.feat_dim,
                                 ds.feat_dtype, xs));
  }
  if (!b->lean) {
    if (node_access_dev_) // Profiler::LogNodeAccess (profiler.cc:570-575): visits per node, counted on the device
      SAM_GGMS(ggms_count_nodes(node_access_dev_, b->input_nodes, max_unique_, n_in, xs));
    SAM_HIP(hipStreamWaitEvent(xs, b->ev_label, 0)); // the batch is complete when its labels are, too
    SAM_HIP(hipMemcpyAsync(b->counts, b->counts_dev, (3 * L + 8) * 8, hipMemcpyDeviceToHost, xs));
    SAM_HIP(hipEventRecord(b->ev_done, xs));
  }
  {
    std::lock_guard<std::mutex> lk(pool_mu_);
    pool_.push_back(b); // graph_pool->Submit
  }
  pool_cv_.notify_all();
  return true;
}

// ---- the host-staged feature path: arch6 without `gpu_extract` (the reference's SGNN mode) --------------------------
// Reference, cache > 0: DoArch6GetCacheMissIndex + DoCacheIdCopyToCPU + DoArch6CacheFeatureCopy (dist_loops.cc:1015-1207:
// split on the GPU, miss ids to the host, ExtractMissData on the CPU, ONE H2D copy, CombineMissData, CombineCacheData --
// every phase behind a StreamSync); cache 0: DoIdCopy + DoCPUFeatureExtract + DoFeatureCopy (dist_loops.cc:481-583,
// dist_loops_arch6.cc:111-133).  Same data flow here, as a pipeline:
//   * the split is enqueued right behind the sampler with the batch size left on the device
//     (ggms_get_miss_cache_index_dev), the hit rows are combined from the cache shards while the host works;
//   * the miss rows go through pinned memory (hipHostMalloc) in CHUNKS: the host team gathers chunk k + 1 while chunk k's
//     asynchronous H2D copy and its scatter into the batch run -- the copy engine, the combine kernel and the cores
//     overlap instead of taking turns, and the last chunks of batch k overlap the first of batch k + 1;
//   * cache 0: every row is a miss and lands where it belongs -- chunks are copied straight into the batch's feature
//     buffer, no split and no combine;
//   * two short host waits per batch (sizes, miss ids) instead of one per phase.
// `staged_serial_epochs` / `staged_serial_steps` (config keys: the first N epochs / batches) or SAMGRAPH_STAGED_SERIAL=1:
// the reference's serial sequence instead, every phase
// timed behind its own wait and logged under the reference's items (kLogL3CacheExtractMissTime ...): the per-phase
// rates of study/host-extract-speed-amount/data.dat are measured this way.

// one row into pinned memory: streaming stores (no read-for-ownership of a buffer the CPU never reads back)
static inline void copy_row_stream(char *dst, const char *src, size_t bytes) {
  typedef long long v2di __attribute__((vector_size(16), aligned(1)));
  typedef long long v2da __attribute__((vector_size(16)));
  size_t i = 0;
  if (((uintptr_t)dst & 15) == 0)
    for (; i + 16 <= bytes; i += 16) __builtin_nontemporal_store(*(const v2di *)(src + i), (v2da *)(dst + i));
  if (i < bytes) std::memcpy(dst + i, src + i, bytes - i);
}

static inline void store_fence() { // streaming stores are weakly ordered: drain them before the DMA engine is told to read
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  asm volatile("sfence" ::: "memory");
#else
  std::atomic_thread_fence(std::memory_order_seq_cst);
#endif
}

void Engine::HostGatherRows(char *rows, const uint32_t *ids, size_t first, size_t count) {
  const char *feat = (const char *)ds.feat.ptr;
  const size_t row_bytes = ds.feat_dim * ggms_dtype_bytes(ds.feat_dtype);
  const uint32_t mask = ds.feat_mask;
  host_team_->ParallelFor(count, [&](size_t lo, size_t hi, int) { // ExtractMissData, cuda_cache_manager_host.cc:268-300
    constexpr size_t kAhead = 8; // rows: a random 512-byte row is 8 cache lines nobody has asked for yet
    for (size_t i = lo; i < hi; ++i) {
      if (i + kAhead < hi) {
        const char *nx = feat + (size_t)(ids[first + i + kAhead] & mask) * row_bytes;
        for (size_t o = 0; o < row_bytes; o += 64) __builtin_prefetch(nx + o, 0, 0);
      }
      copy_row_stream(rows + (first + i) * row_bytes, feat + (size_t)(ids[first + i] & mask) * row_bytes, row_bytes);
    }
    store_fence();
  });
}

void Engine::StagedExtract(Batch *b, hipStream_t ss, hipStream_t xs) {
  const uint32_t L = (uint32_t)cfg.fanout.size();
  const size_t row_bytes = ds.feat_dim * ggms_dtype_bytes(ds.feat_dtype);
  if (!host_team_) {
    host_team_ = std::make_unique<Team>((int)std::max<size_t>(1, cfg.omp_thread_num));
    log_info("staged extract: host team of " + std::to_string(host_team_->size()) + " threads (omp_thread_num)");
  }
  static const bool env_serial = getenv("SAMGRAPH_STAGED_SERIAL") != nullptr;
  const bool serial = env_serial || cur_epoch_ < cfg.staged_serial_epochs || staged_batches_ < cfg.staged_serial_steps;
  ++staged_batches_;
  const bool have_cache = cache_table_ != nullptr;
  uint64_t *n_in = b->counts_dev + 3 * L, *n_miss = b->counts_dev + 3 * L + 2, *n_hit = b->counts_dev + 3 * L + 3;
  using clk = std::chrono::steady_clock;
  auto since = [](clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); };
  double t_index = 0, t_ids = 0, t_gather = 0, t_copy = 0, t_comb_miss = 0, t_comb_hit = 0;
  auto t0 = clk::now();
  // 0. split (GetMissCacheIndex) behind the sampler ON THE BATCH'S SAMPLING STREAM, sizes to the host: wait 1.  The
  // extract stream may still be copying the previous batch's last chunks down -- the cores must not wait for that
  if (have_cache)
    SAM_GGMS(ggms_get_miss_cache_index_dev(cache_table_, b->input_nodes, max_unique_, n_in, b->miss_src, b->miss_dst, n_miss,
                                           b->hit_src, b->hit_dst, n_hit, b->idx_ws,
                                           ggms_cache_index_workspace_bytes(max_unique_), ss));
  SAM_HIP(hipMemcpyAsync(b->counts, b->counts_dev, (3 * L + 8) * 8, hipMemcpyDeviceToHost, ss));
  SAM_HIP(hipStreamSynchronize(ss));
  const size_t num_input = b->counts[3 * L];
  size_t num_miss = have_cache ? b->counts[3 * L + 2] : num_input, num_hit = have_cache ? b->counts[3 * L + 3] : 0;
  if (!have_cache) { // every row comes from the host tier: the counters say so too
    b->counts[3 * L + 2] = num_input;
    SAM_HIP(hipMemcpyAsync(n_miss, b->counts + 3 * L + 2, 8, hipMemcpyHostToDevice, ss));
  }
  SAM_CHECK(num_miss + num_hit == num_input, "CHECK_EQ(num_miss + num_cache, num_input), dist_loops.cc:1047");
  t_index = since(t0);
  if (num_input == 0) return;
  // 1. miss ids to the host (DoCacheIdCopyToCPU / DoIdCopy): wait 2 -- the hit rows are combined meanwhile
  const uint32_t *ids_dev = have_cache ? b->miss_src : b->input_nodes;
  t0 = clk::now();
  if (num_miss) SAM_HIP(hipMemcpyAsync(b->miss_ids_host, ids_dev, num_miss * 4, hipMemcpyDeviceToHost, ss));
  SAM_HIP(hipEventRecord(b->ev_ids, ss));
  SAM_HIP(hipStreamWaitEvent(xs, b->ev_ids, 0)); // the extract stream's work of this batch starts behind the split
  auto combine_hits = [&] { // CombineCacheData
    if (!num_hit) return;
    if (num_cache_part_ == 0)
      SAM_GGMS(ggms_gather_scatter(b->feat, cache_parts_[0], b->hit_src, b->hit_dst, num_hit, nullptr, ds.feat_dim,
                                   ds.feat_dtype, xs));
    else
      SAM_GGMS(ggms_gather_scatter_partition(b->feat, (const void *const *)cache_parts_.data(), num_cache_part_, b->hit_src,
                                             b->hit_dst, num_hit, nullptr, ds.feat_dim, ds.feat_dtype, xs));
  };
  if (!serial) combine_hits(); // on the GPU while the cores gather
  SAM_HIP(hipEventSynchronize(b->ev_ids));
  t_ids = since(t0);
  char *rows = (char *)b->miss_rows_host;
  // where a chunk of miss rows lands on the device: the staging area (scattered by CombineMissData), or -- no cache,
  // rows in batch order -- the batch's feature buffer itself
  char *land = have_cache ? (char *)b->miss_rows_dev : (char *)b->feat;
  if (serial) { // the reference's sequence, one phase at a time
    t0 = clk::now();
    HostGatherRows(rows, b->miss_ids_host, 0, num_miss);
    t_gather = since(t0);
    t0 = clk::now();
    if (num_miss) SAM_HIP(hipMemcpyAsync(land, rows, num_miss * row_bytes, hipMemcpyHostToDevice, xs));
    SAM_HIP(hipStreamSynchronize(xs));
    t_copy = since(t0);
    t0 = clk::now();
    if (have_cache && num_miss)
      SAM_GGMS(ggms_gather_scatter(b->feat, b->miss_rows_dev, nullptr, b->miss_dst, num_miss, nullptr, ds.feat_dim,
                                   ds.feat_dtype, xs)); // CombineMissData
    SAM_HIP(hipStreamSynchronize(xs));
    t_comb_miss = since(t0);
    t0 = clk::now();
    combine_hits();
    SAM_HIP(hipStreamSynchronize(xs));
    t_comb_hit = since(t0);
  } else {
    static const size_t chunk_mb = [] { const char *e = getenv("SAMGRAPH_STAGED_CHUNK_MB"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 16); }();
    static const size_t chunk_rows = [] { const char *e = getenv("SAMGRAPH_STAGED_CHUNK_ROWS"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 0); }(); // test hook
    const size_t chunk = chunk_rows ? chunk_rows : std::max<size_t>(1024, (chunk_mb << 20) / row_bytes);
    // ONE dispatch of the host team per batch: every thread walks the chunks itself (its slice of chunk 0, of chunk 1,
    // ...) and ticks the chunk's counter; the calling thread -- thread 0 of the team, the only one that talks to HIP --
    // hands every chunk whose counter is full to the copy engine between two of its own slices.  (One dispatch per
    // chunk was 34 wake-ups of 15 sleeping threads per batch: a millisecond of an 11-ms step.)
    const size_t nchunks = (num_miss + chunk - 1) / chunk;
    const int T = host_team_->size();
    std::vector<std::atomic<int>> ticks(nchunks);
    for (auto &t : ticks) t.store(0, std::memory_order_relaxed);
    size_t flushed = 0;
    auto flush_ready = [&](bool all) {
      while (flushed < nchunks) {
        if (ticks[flushed].load(std::memory_order_acquire) != T) {
          if (!all) return;
          std::this_thread::yield();
          continue;
        }
        const size_t lo = flushed * chunk, m = std::min(chunk, num_miss - lo);
        SAM_HIP(hipMemcpyAsync(land + lo * row_bytes, rows + lo * row_bytes, m * row_bytes, hipMemcpyHostToDevice, xs));
        if (have_cache)
          SAM_GGMS(ggms_gather_scatter(b->feat, (char *)b->miss_rows_dev + lo * row_bytes, nullptr, b->miss_dst + lo, m, nullptr,
                                       ds.feat_dim, ds.feat_dtype, xs)); // CombineMissData of this chunk
        ++flushed;
      }
    };
    t0 = clk::now();
    const char *feat = (const char *)ds.feat.ptr;
    const uint32_t *ids = b->miss_ids_host;
    const uint32_t mask = ds.feat_mask;
    host_team_->ParallelFor((size_t)T, [&](size_t tid, size_t, int) { // one iteration per thread: iteration == thread
      constexpr size_t kAhead = 8;
      for (size_t c = 0; c < nchunks; ++c) {
        const size_t base = c * chunk, m = std::min(chunk, num_miss - base);
        const size_t q = m / T, r = m % T;
        const size_t lo = base + tid * q + std::min<size_t>(tid, r), hi = lo + q + (tid < r ? 1 : 0);
        for (size_t i = lo; i < hi; ++i) {
          if (i + kAhead < hi) {
            const char *nx = feat + (size_t)(ids[i + kAhead] & mask) * row_bytes;
            for (size_t o = 0; o < row_bytes; o += 64) __builtin_prefetch(nx + o, 0, 0);
          }
          copy_row_stream(rows + i * row_bytes, feat + (size_t)(ids[i] & mask) * row_bytes, row_bytes);
        }
        store_fence();
        ticks[c].fetch_add(1, std::memory_order_release);
        if (tid == 0) flush_ready(false);
      }
    });
    t_gather = since(t0);
    flush_ready(true);
  }
  // the reference's step items (profiler.h:111-116: 44 .. 49); overlapped mode: the host's own busy time per phase
  prof.LogStep(b->key, 44, t_index);
  prof.LogStep(b->key, 45, t_ids);
  prof.LogStep(b->key, 46, t_gather);
  prof.LogStep(b->key, 47, t_copy);
  prof.LogStep(b->key, 48, t_comb_miss);
  prof.LogStep(b->key, 49, t_comb_hit);
  prof.LogEpochAdd(b->key, 20 /*extension: host gather seconds of the staged path*/, t_gather);
  prof.LogEpochAdd(b->key, 21 /*extension: H2D seconds (serial mode)*/, t_copy);
  prof.LogEpochAdd(b->key, 22 /*extension: combine-miss seconds (serial mode)*/, t_comb_miss);
  prof.LogEpochAdd(b->key, 23 /*extension: combine-cache seconds (serial mode)*/, t_comb_hit);
}

// block until the batch is complete, publish sizes, log the items the scripts read
// prev: the batch handed out before this one (still owned by the caller: its timer has not been re-armed)
void Engine::Finish(Batch *b, Batch *prev) {
  if (cfg.arch == kArch0) return; // complete (and logged) when it was enqueued
  SAM_HIP(hipEventSynchronize(b->ev_done));
  double us_gather = 0; // a lean batch's rows: complete when its gather is (the timer's end event)
  double us_busy = 0;   // ... and what it added to the extract streams' busy time
  if (b->lean) {
    SAM_GGMS(ggms_launch_timer_elapsed_us(b->gather_timer, &us_gather));
    us_busy = us_gather;
    // Two extract streams: this gather may have started while the previous batch's was still running.  The epoch's
    // copy time (kLogEpochCopyTime: bytes / time = the extract rate the scripts print) counts every moment once:
    // this batch adds  min(own duration, its end - the previous gather's end).
    if (stream_extract2_ && prev && prev->lean) {
      double us_span = 0, us_prev = 0; // previous start -> this end; previous duration
      if (ggms_launch_timer_span_us(prev->gather_timer, b->gather_timer, &us_span) == GGMS_OK &&
          ggms_launch_timer_elapsed_us(prev->gather_timer, &us_prev) == GGMS_OK)
        us_busy = std::min(us_gather, std::max(0.0, us_span - us_prev));
    }
  }
  const uint32_t L = (uint32_t)cfg.fanout.size();
  b->num_input = b->counts[3 * L];
  b->num_miss = b->counts[3 * L + 2];
  // a kernel of the batch hit a bound it must not hit (include/ggms.h, device status word): the reference
  // CHECK-aborts in these places (logging.cc:69-73), and so does the engine
  if (b->counts[3 * L + 1] != 0) {
    fprintf(stderr, "[samgraph] FATAL: device status %#llx after batch %llu (%s%s): results are invalid\n",
            (unsigned long long)b->counts[3 * L + 1], (unsigned long long)b->key,
            (b->counts[3 * L + 1] & GGMS_STATUS_SCAN_SPIN) ? "ordered scan: look-back gave up " : "",
            (b->counts[3 * L + 1] & GGMS_STATUS_TABLE_FULL) ? "hashed dedup table full" : "");
    abort();
  }
  float ms_sample = 0, ms_copy = 0;
  (void)hipEventElapsedTime(&ms_sample, b->ev_start, b->ev_sampled);
  if (b->lean) ms_copy = (float)(us_gather * 1e-3); // the gather kernel's own time
  else (void)hipEventElapsedTime(&ms_copy, b->ev_xstart, b->ev_done); // not from ev_sampled: that would add the queueing behind the previous extract
  const double s_copy_epoch = b->lean ? us_busy * 1e-6 : ms_copy * 1e-3;
  uint64_t edges = 0;
  for (uint32_t i = 0; i < L; ++i) edges += b->counts[3 * i];
  const double row_bytes = (double)ds.feat_dim * ggms_dtype_bytes(ds.feat_dtype);
  // item codes: profiler.h:58-140 (kLogL1NumSample = 0, kLogL1NumNode = 1, kLogL1SampleTime = 3,
  // kLogL1CopyTime = 6, kLogL1FeatureBytes = 9, kLogL1MissBytes = 13; epoch items :119-137)
  prof.LogStep(b->key, 0, (double)edges);
  prof.LogStep(b->key, 1, (double)b->num_input);
  prof.LogStep(b->key, 3, ms_sample * 1e-3);
  prof.LogStep(b->key, 6, ms_copy * 1e-3);
  prof.LogStep(b->key, 9, b->num_input * row_bytes);
  prof.LogStep(b->key, 13, b->num_miss * row_bytes);
  prof.LogEpochAdd(b->key, 0 /*kLogEpochSampleTime*/, ms_sample * 1e-3);
  prof.LogEpochAdd(b->key, 8 /*kLogEpochCopyTime*/, s_copy_epoch);
  prof.LogEpochAdd(b->key, 12 /*kLogEpochFeatureBytes*/, b->num_input * row_bytes);
  prof.LogEpochAdd(b->key, 13 /*kLogEpochMissBytes*/, b->num_miss * row_bytes);
  prof.LogEpochAdd(b->key, 15 /*kLogEpochNumSample*/, (double)edges);
}

uint64_t Engine::GetNextBatch() { // operation.cc:366-378 + GraphPool::GetGraphBatch graph_pool.cc:31-49
  // the engine drops its own reference to the previous batch (:370) -- once the next one's gather has been measured
  // against it (Finish): until then its slot, and with it its launch timer, is not handed to the sampler again
  Batch *prev = current_;
  current_ = nullptr;
  Batch *b = nullptr;
  {
    std::unique_lock<std::mutex> lk(pool_mu_);
    if (pool_.empty() && !bg_running_.load())
      fatal(__FILE__, __LINE__, "get_next_batch with nothing sampled: call sample_once() or extract_start() first");
    pool_cv_.wait(lk, [&] { return !pool_.empty() || bg_stop_.load(); });
    if (pool_.empty()) fatal(__FILE__, __LINE__, "engine shut down while waiting for a batch");
    b = pool_.front();
    pool_.pop_front();
  }
  Finish(b, prev);
  if (prev) {
    std::lock_guard<std::mutex> lk(pool_mu_);
    prev->in_use = false;
  }
  current_ = b;
  return b->key;
}

void Engine::ExtractStart(int count) { // dist_engine.cc StartExtract: one background sample+extract thread
  (void)count;
  SAM_CHECK(!bg_running_.load(), "extract thread already running");
  bg_running_ = true; // set before the thread exists: nothing the thread runs looks at bg_ itself
  bg_ = std::thread([this] {
    SAM_HIP(hipSetDevice(device_));
    while (!bg_stop_) {
      const size_t before_epoch = cur_epoch_;
      RunSampleOnce(true);
      if (cur_epoch_ >= cfg.num_epoch && before_epoch >= cfg.num_epoch) break;
      if (cur_epoch_ >= cfg.num_epoch) break;
    }
  });
}

// samgraph_report_node_access, the _SIMPLE report (profiler.cc:795-860): nodes by visit count, descending --
//   node_access_optimal_cache_bin<t>.txt       u32 node ids in that order (usable as a cache rank file)
//   node_access_optimal_cache_freq_bin<t>.txt  f32 visits per epoch, same order
//   node_access_frequency<t>.txt               "rate\tvisits per epoch of the node at that percentile"
//   node_access_optimal_cache_hit<t>.txt       "rate\thit rate of a cache holding the top rate % of the nodes"
void Engine::ReportNodeAccess() {
  if (!node_access_dev_) return;
  SAM_HIP(hipDeviceSynchronize());
  std::vector<uint32_t> freq(ds.num_node);
  SAM_HIP(hipMemcpy(freq.data(), node_access_dev_, ds.num_node * 4, hipMemcpyDeviceToHost));
  std::vector<std::pair<uint64_t, uint32_t>> rec(ds.num_node);
  for (uint32_t v = 0; v < ds.num_node; ++v) rec[v] = {freq[v], v};
  std::sort(rec.begin(), rec.end(), std::greater<std::pair<uint64_t, uint32_t>>());
  const std::string t = std::to_string((unsigned long long)std::chrono::system_clock::now().time_since_epoch().count());
  FILE *f_freq = fopen(("node_access_frequency" + t + ".txt").c_str(), "w");
  FILE *f_bin = fopen(("node_access_optimal_cache_bin" + t + ".txt").c_str(), "wb");
  FILE *f_hit = fopen(("node_access_optimal_cache_hit" + t + ".txt").c_str(), "w");
  FILE *f_fbin = fopen(("node_access_optimal_cache_freq_bin" + t + ".txt").c_str(), "wb");
  SAM_CHECK(f_freq && f_bin && f_hit && f_fbin, "cannot write the node access files");
  const double epochs = (double)std::max<size_t>(1, cfg.num_epoch);
  uint64_t sum = 0;
  for (auto &p : rec) {
    const float avg = (float)(p.first / epochs);
    fwrite(&p.second, 4, 1, f_bin);
    fwrite(&avg, 4, 1, f_fbin);
    sum += p.first;
    p.first = sum; // running sum, as the reference keeps it
  }
  const size_t n = rec.size();
  for (int rate = 0; rate <= 100; ++rate) {
    size_t idx = rate == 0 ? 0 : ((uint64_t)rate * n - 1) / 100;
    fprintf(f_freq, "%d\t%g\n", rate, idx == 0 ? 0.0 : (double)(rec[idx].first - rec[idx - 1].first) / epochs);
    fprintf(f_hit, "%d\t%g\n", rate, idx == 0 || sum == 0 ? 0.0 : (double)rec[idx].first / (double)sum);
  }
  fclose(f_freq); fclose(f_bin); fclose(f_hit); fclose(f_fbin);
}

Batch *Engine::Current(uint64_t key) {
  SAM_CHECK(current_ != nullptr, "no current batch");
  SAM_CHECK(current_->key == key, "key is not the current batch key (adapter.cc:68)");
  return current_;
}

void Engine::Retain(uint64_t key) { Current(key)->refs.fetch_add(1); }

// batch keys are unique over a run (epoch * steps + step), and a slot is only reused once its refs are back to 0,
// so a key names at most one live slot
void Engine::Release(uint64_t key) {
  for (auto &b : slots_)
    if (b->key == key && b->refs.load() > 0) { b->refs.fetch_sub(1); return; }
}

} // namespace sam
