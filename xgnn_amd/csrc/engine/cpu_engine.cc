// cpu_engine.cc -- arch0: sampler and extractor on the host cores, trainer on a GPU (or on the host).
//
// BASELINE configs[0] ("CPU arch, 1 worker -- plumbing").  Counterpart of the reference's CPUEngine
// (cpu/cpu_engine.cc, cpu/cpu_loops.cc:41-299, cpu/cpu_loops_arch0.cc:161-167) with its leaves restated in this
// file: CPUShuffler (cpu/cpu_shuffler.cc:32-111), CPUSampleKHop0 / CPUSampleKHop2 (cpu/cpu_sampling_khop{0,2}.cc),
// RandomID (cpu/cpu_random.cc:26-30), CPUHashTable2 (cpu/cpu_hashtable2.cc:53-191, the reference's default table,
// run_config.cc:56) and CPUExtract (cpu/cpu_extraction.cc:31-90).
//
// This is a deployment of its own, selected by `_arch = 0` -- NOT a fallback of the GPU path: arch1 / arch6 never
// come here, and without a GPU they fail loudly.  Nothing in this file touches oracle/.
//
// Parallelism follows the reference's OpenMP loops: a persistent team of `omp_thread_num` threads, thread 0 being
// the caller, every `parallel for` split into one contiguous block per thread (OpenMP's default static schedule),
// each thread with its own default-seeded std::mt19937 that lives as long as the thread (`static thread_local` in
// RandomID).  With one thread the results are those of the reference at omp_thread_num = 1, draw for draw.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <random>

#include "engine.h"
#include "team.h"

namespace sam {

namespace {

// RandomID (cpu_random.cc:26-30): inclusive range, one generator per thread, default seed, never re-seeded
uint32_t RandomId(uint32_t lo, uint32_t hi) {
  static thread_local std::mt19937 generator;
  std::uniform_int_distribution<uint32_t> d(lo, hi);
  return d(generator);
}

constexpr uint32_t kEmpty = GGMS_EMPTY_KEY;

// keep what is not kEmptyKey, in order (the reference's single-thread std::remove_if, cpu_sampling_khop0.cc:72-79)
size_t Compact(uint32_t *a, size_t n) {
  size_t w = 0;
  for (size_t i = 0; i < n; ++i)
    if (a[i] != kEmpty) a[w++] = a[i];
  return w;
}

// CPUSampleKHop0 (reservoir, positions >= fanout draw from [0, j + 1] INCLUSIVE) and CPUSampleKHop2 (partial
// Fisher-Yates that permutes `indices` in place); padded per seed, then compacted
size_t SampleKHop(Team &team, bool khop2, const uint32_t *indptr, uint32_t *indices, const uint32_t *input, size_t n,
                  size_t fanout, uint32_t *src, uint32_t *dst) {
  std::vector<char> short_list(team.size(), 0);
  team.ParallelFor(n, [&](size_t lo, size_t hi, int tid) {
    for (size_t i = lo; i < hi; ++i) {
      const uint32_t rid = input[i], off = indptr[rid], len = indptr[rid + 1] - off;
      uint32_t *s = src + i * fanout, *d = dst + i * fanout;
      if (len < fanout) short_list[tid] = 1;
      if (len <= fanout) {
        for (size_t j = 0; j < fanout; ++j) {
          s[j] = j < len ? rid : kEmpty;
          d[j] = j < len ? indices[off + j] : kEmpty;
        }
      } else if (!khop2) {
        for (size_t j = 0; j < fanout; ++j) {
          s[j] = rid;
          d[j] = indices[off + j];
        }
        for (uint32_t j = (uint32_t)fanout; j < len; ++j) {
          const uint32_t k = RandomId(0, j + 1);
          if (k < fanout) d[k] = indices[off + j];
        }
      } else {
        for (uint32_t j = 0; j < fanout; ++j) {
          const uint32_t k = RandomId(0, len - j - 1);
          s[j] = rid;
          d[j] = indices[off + k];
          std::swap(indices[off + k], indices[off + len - j - 1]);
        }
      }
    }
  });
  if (std::none_of(short_list.begin(), short_list.end(), [](char c) { return c != 0; })) return n * fanout;
  const size_t m = Compact(src, n * fanout);
  Compact(dst, n * fanout);
  return m;
}

// CPUHashTable2: one bucket per node id; Populate = claim by CAS, count per thread, prefix over the threads, number
// per thread -- ids follow input order inside a thread's block and the blocks are in input order
class NodeTable {
 public:
  NodeTable(size_t num_node, Team &team) : team_(team), o2n_(num_node), n2o_(num_node) {
    team_.ParallelFor(num_node, [&](size_t lo, size_t hi, int) {
      for (size_t i = lo; i < hi; ++i) o2n_[i] = Bucket{kEmpty, kEmpty, kEmpty, kEmpty};
    });
  }
  void Reset() { // :183-191
    team_.ParallelFor(num_items_, [&](size_t lo, size_t hi, int) {
      for (size_t i = lo; i < hi; ++i) o2n_[n2o_[i]].key = kEmpty;
    });
    num_items_ = 0;
    version_ = 0;
  }
  void Populate(const uint32_t *input, size_t n) { // :53-107
    const uint32_t version = version_;
    team_.ParallelFor(n, [&](size_t lo, size_t hi, int) {
      for (size_t i = lo; i < hi; ++i) {
        Bucket &b = o2n_[input[i]];
        if (__sync_val_compare_and_swap(&b.key, kEmpty, input[i]) == kEmpty) {
          b.index = (uint32_t)i;
          b.version = version;
        }
      }
    });
    std::vector<size_t> first(team_.size() + 1, 0);
    team_.ParallelFor(n, [&](size_t lo, size_t hi, int tid) {
      size_t c = 0;
      for (size_t i = lo; i < hi; ++i) {
        const Bucket &b = o2n_[input[i]];
        c += (b.index == (uint32_t)i && b.version == version);
      }
      first[tid + 1] = c;
    });
    for (int t = 0; t < team_.size(); ++t) first[t + 1] += first[t];
    const size_t start = num_items_;
    team_.ParallelFor(n, [&](size_t lo, size_t hi, int tid) {
      size_t next = start + first[tid];
      for (size_t i = lo; i < hi; ++i) {
        Bucket &b = o2n_[input[i]];
        if (b.index == (uint32_t)i && b.version == version) {
          b.local = (uint32_t)next;
          n2o_[next++] = input[i];
        }
      }
    });
    num_items_ += first[team_.size()];
    ++version_;
  }
  size_t NumItems() const { return num_items_; }
  const uint32_t *Unique() const { return n2o_.data(); } // MapNodes, :142-146
  void MapEdges(const uint32_t *src, const uint32_t *dst, size_t n, uint32_t *new_src, uint32_t *new_dst) { // :148-160
    team_.ParallelFor(n, [&](size_t lo, size_t hi, int) {
      for (size_t i = lo; i < hi; ++i) {
        new_src[i] = o2n_[src[i]].local;
        new_dst[i] = o2n_[dst[i]].local;
      }
    });
  }

 private:
  struct Bucket { uint32_t key, local, index, version; };
  Team &team_;
  std::vector<Bucket> o2n_;
  std::vector<uint32_t> n2o_;
  size_t num_items_ = 0;
  uint32_t version_ = 0;
};

// CPUExtract: out[i, :] = src[index[i], :], rows of any element type as bytes
// mask: CPUMockExtract (cpu_extraction.cc:47-62), the 2^k-row stand-in table of SAMGRAPH_EMPTY_FEAT
void Extract(Team &team, void *dst, const void *src, const uint32_t *index, size_t n, size_t row_bytes,
             uint32_t mask = 0xffffffffu) {
  team.ParallelFor(n, [&](size_t lo, size_t hi, int) {
    for (size_t i = lo; i < hi; ++i)
      std::memcpy((char *)dst + i * row_bytes, (const char *)src + (size_t)(index[i] & mask) * row_bytes, row_bytes);
  });
}

double Seconds(std::chrono::steady_clock::time_point a) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
}

} // namespace

// everything arch0 owns besides the batch slots
struct Engine::CpuPath {
  explicit CpuPath(int threads) : team(threads) {}
  Team team;
  std::unique_ptr<NodeTable> table;
  std::vector<uint32_t> train;      // shuffled in place every epoch (CPUShuffler keeps the tensor it was given)
  std::vector<uint32_t> tmp_src, tmp_dst, cur;
  size_t num_step = 0, cur_epoch = 0, cur_step = 0;
  bool initialized = false;
  // host staging of a batch when the trainer is a GPU (DoGraphCopy / DoFeatureCopy, cpu_loops.cc:230-299)
  struct Staging {
    std::vector<std::vector<uint32_t>> row, col;
    std::vector<uint32_t> input_nodes, output_nodes;
    std::vector<char> feat;
    std::vector<int64_t> label;
  };
  std::vector<Staging> staging; // one per slot
};

// defined where CpuPath is complete (the engine holds it through a unique_ptr)
Engine::Engine() = default;
Engine::~Engine() = default;

void Engine::CpuInit() { // CPUEngine::Init, cpu_engine.cc:50-110
  SAM_CHECK(cfg.sample_type == GGMS_KHOP0 || cfg.sample_type == GGMS_KHOP2,
            "arch0 samples with khop0 or khop2 (cpu_loops.cc:98-109)");
  DataInit();
  cpu_ = std::make_unique<CpuPath>((int)cfg.omp_thread_num);
  CpuPath &C = *cpu_;
  C.table = std::make_unique<NodeTable>(ds.num_node, C.team);
  const uint32_t *train = (const uint32_t *)ds.train_set.ptr;
  C.train.assign(train, train + ds.num_train);
  C.num_step = (ds.num_train + cfg.batch_size - 1) / cfg.batch_size; // drop_last = false
  C.cur_step = C.num_step;
  num_local_step_ = num_global_step_ = C.num_step;
  const uint32_t L = (uint32_t)cfg.fanout.size();
  max_seeds_ = cfg.batch_size;
  max_input_.resize(L);
  max_edges_.resize(L);
  SAM_GGMS(ggms_sample_batch_capacity(max_seeds_, cfg.fanout.data(), L, max_input_.data(), max_edges_.data(), &max_unique_));
  size_t e_all = 0;
  for (auto e : max_edges_) e_all = std::max(e_all, e);
  C.tmp_src.resize(e_all + 1);
  C.tmp_dst.resize(e_all + 1);
  if (cfg.sample_type == GGMS_KHOP2 && !ds.indices.owned) { // khop2 permutes the neighbour lists: a private copy
    void *copy = std::malloc(std::max<size_t>(ds.indices.bytes, 4));
    std::memcpy(copy, ds.indices.ptr, ds.indices.bytes);
    ds.indices.ptr = copy;
    ds.indices.owned = true;
    ds.indices.mapped_file = ds.indices.shared_anon = false;
  }
  const bool gpu = !cfg.trainer_on_host;
  device_ = cfg.trainer_device;
  if (gpu) {
    SAM_HIP(hipSetDevice(device_));
    SAM_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  }
  const size_t row_bytes = ds.feat_dim * ggms_dtype_bytes(ds.feat_dtype);
  size_t nslots = 2;
  if (cfg.raw.count("max_copying_jobs")) nslots = std::max<size_t>(2, std::min<size_t>(4, std::stoull(cfg.raw["max_copying_jobs"]) + 1));
  nslots = std::max(nslots, 2 + cfg.lookahead);
  C.staging.resize(nslots);
  auto alloc = [&](size_t bytes) -> void * {
    void *p = nullptr;
    if (gpu) SAM_HIP(hipMalloc(&p, std::max<size_t>(bytes, 16)));
    else p = std::malloc(std::max<size_t>(bytes, 16));
    SAM_CHECK(p != nullptr, "out of memory");
    return p;
  };
  for (size_t s = 0; s < nslots; ++s) {
    auto b = std::make_unique<Batch>();
    b->slot = (int)s;
    b->host = !gpu;
    b->row.resize(L); b->col.resize(L); b->data.resize(L, nullptr);
    for (uint32_t i = 0; i < L; ++i) {
      b->row[i] = (uint32_t *)alloc(max_edges_[i] * 4);
      b->col[i] = (uint32_t *)alloc(max_edges_[i] * 4);
    }
    b->input_nodes = (uint32_t *)alloc(max_unique_ * 4);
    b->output_nodes = (uint32_t *)alloc(max_seeds_ * 4);
    b->feat = alloc(max_unique_ * row_bytes);
    b->label = (int64_t *)alloc(max_seeds_ * 8);
    b->counts = (uint64_t *)std::calloc(3 * L + 8, 8);
    if (gpu) {
      auto &S = C.staging[s];
      S.row.resize(L); S.col.resize(L);
      for (uint32_t i = 0; i < L; ++i) { S.row[i].resize(max_edges_[i] + 1); S.col[i].resize(max_edges_[i] + 1); }
      S.input_nodes.resize(max_unique_ + 1);
      S.output_nodes.resize(max_seeds_ + 1);
      S.feat.resize(max_unique_ * row_bytes + 16);
      S.label.resize(max_seeds_ + 1);
    }
    slots_.push_back(std::move(b));
  }
  prof.Resize(cfg.num_epoch, num_global_step_);
  sample_ready_ = train_ready_ = true;
}

// RunArch0LoopsOnce (cpu_loops_arch0.cc:161-167): DoShuffle, DoCPUSample, DoFeatureExtract, DoGraphCopy,
// DoFeatureCopy -- synchronous, the batch is complete when this returns
bool Engine::CpuEnqueueOne(bool background) {
  SAM_CHECK(train_ready_, "engine not initialised");
  CpuPath &C = *cpu_;
  Batch *b = AcquireSlot(background);
  if (!b) return false;
  // ---- CPUShuffler::GetBatch / ReShuffle (cpu_shuffler.cc:54-111)
  ++C.cur_step;
  if (C.cur_step >= C.num_step) {
    if (!C.initialized) { C.cur_epoch = 0; C.initialized = true; } else { ++C.cur_epoch; }
    C.cur_step = 0;
    if (C.cur_epoch < cfg.num_epoch) {
      const uint64_t seed = cfg.has_seed ? cfg.seed + C.cur_epoch
                                         : (uint64_t)std::chrono::system_clock::now().time_since_epoch().count(); // :68
      auto g = std::default_random_engine(seed);
      const size_t n = C.train.size();
      for (size_t i = 0; n && i < n - 1; i++) {
        std::uniform_int_distribution<size_t> d(i, n - 1);
        std::swap(C.train[i], C.train[d(g)]);
      }
    }
  }
  cur_epoch_ = C.cur_epoch;
  if (C.cur_epoch >= cfg.num_epoch) { // training finished
    std::lock_guard<std::mutex> lk(pool_mu_);
    b->in_use = false;
    return false;
  }
  const bool gpu = !b->host;
  CpuPath::Staging *S = gpu ? &C.staging[b->slot] : nullptr;
  const size_t offset = C.cur_step * cfg.batch_size;
  const size_t num_seeds = std::min(cfg.batch_size, C.train.size() - offset);
  b->num_seeds = num_seeds;
  b->key = BatchKey(C.cur_epoch, C.cur_step);
  uint32_t *seeds = gpu ? S->output_nodes.data() : b->output_nodes;
  std::memcpy(seeds, C.train.data() + offset, num_seeds * 4);

  // ---- DoCPUSample (cpu_loops.cc:55-192)
  const auto t_sample = std::chrono::steady_clock::now();
  const uint32_t L = (uint32_t)cfg.fanout.size();
  const uint32_t *indptr = (const uint32_t *)ds.indptr.ptr;
  uint32_t *indices = (uint32_t *)ds.indices.ptr;
  C.table->Reset();
  C.table->Populate(seeds, num_seeds);
  C.cur.assign(seeds, seeds + num_seeds);
  uint64_t edges = 0;
  for (int i = (int)L - 1; i >= 0; --i) {
    const size_t n_in = C.cur.size();
    const size_t n_out = SampleKHop(C.team, cfg.sample_type == GGMS_KHOP2, indptr, indices, C.cur.data(), n_in,
                                    cfg.fanout[i], C.tmp_src.data(), C.tmp_dst.data());
    C.table->Populate(C.tmp_dst.data(), n_out);
    const size_t num_unique = C.table->NumItems();
    uint32_t *row = gpu ? S->row[i].data() : b->row[i], *col = gpu ? S->col[i].data() : b->col[i];
    C.table->MapEdges(C.tmp_src.data(), C.tmp_dst.data(), n_out, col, row); // row = new_dst, col = new_src (:151-160)
    b->counts[3 * i + 0] = n_out;
    b->counts[3 * i + 1] = num_unique;
    b->counts[3 * i + 2] = n_in;
    edges += n_out;
    C.cur.assign(C.table->Unique(), C.table->Unique() + num_unique);
  }
  b->num_input = C.cur.size();
  b->counts[3 * L] = b->num_input;
  b->counts[3 * L + 1] = 0;
  b->counts[3 * L + 2] = 0;
  uint32_t *input_nodes = gpu ? S->input_nodes.data() : b->input_nodes;
  std::memcpy(input_nodes, C.cur.data(), b->num_input * 4);
  const double sample_s = Seconds(t_sample);

  // ---- DoFeatureExtract (:194-228) + DoGraphCopy / DoFeatureCopy (:230-299)
  const auto t_copy = std::chrono::steady_clock::now();
  const size_t row_bytes = ds.feat_dim * ggms_dtype_bytes(ds.feat_dtype);
  void *feat = gpu ? (void *)S->feat.data() : b->feat;
  int64_t *label = gpu ? S->label.data() : b->label;
  Extract(C.team, feat, ds.feat.ptr, input_nodes, b->num_input, row_bytes, ds.feat_mask);
  Extract(C.team, label, ds.label.ptr, seeds, num_seeds, 8);
  if (gpu) {
    SAM_HIP(hipSetDevice(device_));
    for (uint32_t i = 0; i < L; ++i) {
      SAM_HIP(hipMemcpyAsync(b->row[i], S->row[i].data(), b->counts[3 * i] * 4, hipMemcpyHostToDevice, stream_));
      SAM_HIP(hipMemcpyAsync(b->col[i], S->col[i].data(), b->counts[3 * i] * 4, hipMemcpyHostToDevice, stream_));
    }
    SAM_HIP(hipMemcpyAsync(b->input_nodes, input_nodes, b->num_input * 4, hipMemcpyHostToDevice, stream_));
    SAM_HIP(hipMemcpyAsync(b->output_nodes, seeds, num_seeds * 4, hipMemcpyHostToDevice, stream_));
    SAM_HIP(hipMemcpyAsync(b->feat, feat, b->num_input * row_bytes, hipMemcpyHostToDevice, stream_));
    SAM_HIP(hipMemcpyAsync(b->label, label, num_seeds * 8, hipMemcpyHostToDevice, stream_));
    SAM_HIP(hipStreamSynchronize(stream_));
  }
  const double copy_s = Seconds(t_copy);
  // the items the scripts read (profiler.h:58-140), as Engine::Finish logs them for the GPU engines
  const double feat_bytes = (double)b->num_input * row_bytes;
  prof.LogStep(b->key, 0, (double)edges);
  prof.LogStep(b->key, 1, (double)b->num_input);
  prof.LogStep(b->key, 3, sample_s);
  prof.LogStep(b->key, 6, copy_s);
  prof.LogStep(b->key, 9, feat_bytes);
  prof.LogStep(b->key, 13, feat_bytes);
  prof.LogEpochAdd(b->key, 0, sample_s);
  prof.LogEpochAdd(b->key, 8, copy_s);
  prof.LogEpochAdd(b->key, 12, feat_bytes);
  prof.LogEpochAdd(b->key, 13, feat_bytes); // kLogEpochMissBytes: every row comes from the host (cpu_loops.cc:298)
  prof.LogEpochAdd(b->key, 15, (double)edges);
  {
    std::lock_guard<std::mutex> lk(pool_mu_);
    pool_.push_back(b);
  }
  pool_cv_.notify_all();
  return true;
}

void Engine::CpuShutdown() { cpu_.reset(); }

} // namespace sam
