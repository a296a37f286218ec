// extract.hip -- feature row gather: GPUExtract and the cache-combine family.
//
// Reference kernels: gpu_extract (cuda/cuda_extraction.cu:31-49), combine_miss_data,
// extract_miss_data, combine_cache_data, combine_cache_data_for_partition
// (cuda/cuda_cache_manager_device.cu:209-299).  All of them are
//     out[dst(i), :] = SRC(src(i))[ : ]
// and differ only in how a source row is located.  One kernel template does all
// of them; the row locator is a functor evaluated once per row.
//
// MI355X mapping (HBM-bound, no reuse, rows of 8 B .. a few KiB):
//   * one wave owns 64 consecutive output rows; each lane resolves ONE row
//     (index load, optional slot -> shard translation) and keeps its source
//     pointer in registers;
//   * the tile is then swept as a flat array of 16-byte chunks: chunk c of the
//     tile belongs to row c / chunks_per_row; the owning lane's pointer comes
//     through ds_bpermute (no LDS memory), so every wave-instruction moves a full
//     1 KiB even when a row is 400 B (25 chunks), and the store side of an
//     identity-destination gather is one contiguous 1 KiB segment;
//   * U = 16 independent chunk loads per lane are issued before the first store.
// Algorithmic bytes per row: 4 (index) + 2 * row_bytes.
#include <cstdlib>

#include <hip/hip_ext.h>

#include "ggms_device.h"

namespace ggms {

template <int BYTES> struct ChunkT;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <> struct ChunkT<16> { using type = u32x4_t; };
template <> struct ChunkT<8> { using type = u32x2_t; };
template <> struct ChunkT<4> { using type = uint32_t; };
template <> struct ChunkT<2> { using type = uint16_t; };
template <> struct ChunkT<1> { using type = uint8_t; };

// which tier served a row (0 = not counted); counters[tier - 1] in ggms_extract_tiered
constexpr uint32_t kTierHost = 1, kTierRemote = 2, kTierLocal = 3, kTierReplica = 4;

// ---- row locators -----------------------------------------------------------
// src(i) = index ? index[i] : i ;  row pointer = base + src(i) * row_bytes
struct PlainRows {
  const char *base;
  const uint32_t *index;
  uint64_t row_bytes;
  uint32_t mask = 0xffffffffu; // gpu_mock_extract (cuda_extraction.cu:51-70): rows of a 2^k-row mock table, index & mask
  static constexpr bool kTiers = false;
  __device__ __forceinline__ const char *row(uint64_t i, uint32_t &tier) const {
    tier = 0;
    const uint64_t s = index ? (uint64_t)(index[i] & mask) : i;
    return base + s * row_bytes;
  }
};

// DeviceDistFeature::Get (cuda/dist_graph.h:191-205): slot -> shard slot % P, row slot / P
// The shard pointers travel by value (PtrSet: a select chain on registers, not a load from a device pointer table) and
// the split is a multiply-high by the launch-constant shard count (Divisor) -- both sit on every row's
// index -> slot -> pointer chain, ahead of the row's first byte.
typedef PtrSet<const char, kMaxParts> PartPtrs;
struct PartitionRows {
  PartPtrs parts;
  const uint32_t *index;
  uint64_t row_bytes;
  Divisor num_part;
  static constexpr bool kTiers = false;
  __device__ __forceinline__ const char *row(uint64_t i, uint32_t &tier) const {
    tier = 0;
    uint32_t part, real;
    num_part.divmod(index[i], real, part);
    return parts.pick(part) + (uint64_t)real * row_bytes;
  }
};

// fused hit/miss: table[node] == kEmptyKey -> host tier row `node`, else cache slot
struct CachedRows {
  PartPtrs parts;
  const uint32_t *nodes;
  const uint32_t *table;
  const char *host;
  uint64_t row_bytes;
  Divisor num_part; // d = 1: one cache array parts[0]
  static constexpr bool kTiers = false;
  __device__ __forceinline__ const char *row(uint64_t i, uint32_t &tier) const {
    const uint32_t node = nodes[i];
    const uint32_t slot = table[node];
    tier = (slot == kEmptyKey) ? kTierHost : 0u;
    if (slot == kEmptyKey) return host + (uint64_t)node * row_bytes;
    uint32_t part, real;
    num_part.divmod(slot, real, part);
    return parts.pick(part) + (uint64_t)real * row_bytes;
  }
};

// All tiers of the store in one locator (ggms_extract_tiered): slot = table ? table[node] : node;
//   kEmptyKey            -> host tier, row `node` of the device-mapped host table      (GPUExtractMissData)
//   slot <  num_replica  -> this GPU's replica of the hottest rows, row `slot`         (hot-row replication)
//   else s = slot - num_replica -> shard s % P at row s / P: local HBM or a peer's over xGMI
//                                                                      (combine_cache_data_for_partition)
// and says which tier served the row, for the per-tier counters (count_local_cache, :171-207).
struct TieredRows {
  PartPtrs parts;
  const uint32_t *nodes;
  const uint32_t *table;
  const char *host;
  const char *replica;
  uint64_t row_bytes;
  uint32_t num_replica;
  Divisor num_part; // d >= 1
  uint32_t my_part;
  uint32_t host_mask; // host tier row = node & host_mask (mock table of SAMGRAPH_EMPTY_FEAT; else all ones)
  static constexpr bool kTiers = true;
  __device__ __forceinline__ const char *row(uint64_t i, uint32_t &tier) const {
    const uint32_t node = nodes[i];
    const uint32_t slot = table ? table[node] : node;
    if (slot == kEmptyKey) {
      tier = kTierHost;
      return host + (uint64_t)(node & host_mask) * row_bytes;
    }
    if (slot < num_replica) {
      tier = kTierReplica;
      return replica + (uint64_t)slot * row_bytes;
    }
    uint32_t part, real;
    num_part.divmod(slot - num_replica, real, part);
    tier = part == my_part ? kTierLocal : kTierRemote;
    return parts.pick(part) + (uint64_t)real * row_bytes;
  }
};

// cache_ratio 1.0, rows kept in NODE order (slot = node id): no table, no miss tier.  The layout of a full cache is
// not observable through the reference's interface (the batch's rows come out in input-node order either way), and
// it removes one dependent random 4-byte read (a 64-byte sector of HBM traffic) per gathered row.
struct IdentRows {
  PartPtrs parts;
  const uint32_t *nodes;
  uint64_t row_bytes;
  Divisor num_part; // d = 1: one array parts[0]
  static constexpr bool kTiers = false;
  __device__ __forceinline__ const char *row(uint64_t i, uint32_t &tier) const {
    tier = 0;
    const uint32_t node = nodes[i];
    if (num_part.d == 1) return parts.p[0] + (uint64_t)node * row_bytes; // uniform
    uint32_t part, real;
    num_part.divmod(node, real, part);
    return parts.pick(part) + (uint64_t)real * row_bytes;
  }
};

// host side: the caller's HOST array of shard base pointers -> kernel argument
static inline bool part_ptrs(const void *const *parts, uint32_t num_part, PartPtrs &out) {
  out = PartPtrs{};
  const uint32_t n = num_part ? num_part : 1;
  if (!parts || n > kMaxParts) {
    set_error("extract: num_part %u (at most %u shards; `parts` is a HOST array of num_part device pointers)", num_part, kMaxParts);
    return false;
  }
  if (!host_readable_table(parts, "`parts`")) return false;
  for (uint32_t p = 0; p < n; ++p) out.p[p] = (const char *)parts[p];
  return true;
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  const uint32_t lo = __shfl((uint32_t)v, src, 64);
  const uint32_t hi = __shfl((uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}

// chunk loads: plain or non-temporal (rows of a batch are read once)
// The pointers travel through ds_bpermute as integers; tell the compiler they are GLOBAL so it emits
// global_load/global_store (vmcnt only) instead of flat_* (vmcnt + lgkmcnt, aperture check).
template <typename V, bool NT>
__device__ __forceinline__ V load_chunk(uint64_t addr) {
  typedef const V __attribute__((address_space(1))) *gp_t;
  gp_t p = (gp_t)addr;
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <typename V, bool NT>
__device__ __forceinline__ void store_chunk(uint64_t addr, V v) {
  typedef V __attribute__((address_space(1))) *gp_t;
  if constexpr (NT) __builtin_nontemporal_store(v, (gp_t)addr);
  else *(gp_t)addr = v;
}

// U = independent chunk loads in flight per lane (16, or 8 for rows of fewer than 8 chunks)
// Loads and stores are non-temporal: a batch's rows are read once and the gathered batch is a > 100-MB stream that
// nothing re-reads from cache (measured + 3..5 % on MI355X each, profiles/r01-r02).
template <int CB, typename Rows, bool IDENT_DST, int U>
__global__ __launch_bounds__(kBlock) void k_gather_rows(char *__restrict__ out, Rows rows,
                                                        const uint32_t *__restrict__ dst_index, Count n_arg,
                                                        uint32_t rc, uint32_t magic, uint64_t *miss_count) {
  using V = typename ChunkT<CB>::type;
  const uint64_t n = n_arg.get();
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const uint64_t num_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  const uint64_t row_bytes = (uint64_t)rc * CB;
  const uint64_t num_tiles = (n + kWave - 1) / kWave;

  // resolve one row per lane for tile `t`: source pointer (+ destination pointer)
  auto resolve = [&](uint64_t t, uint64_t &sp, uint64_t &dp, uint32_t &miss) {
    const uint64_t my_row = t * kWave + lane;
    sp = 0; dp = 0; miss = 0;
    if (t < num_tiles && my_row < n) {
      sp = (uint64_t)rows.row(my_row, miss);
      if constexpr (!IDENT_DST) dp = (uint64_t)(out + (uint64_t)dst_index[my_row] * row_bytes);
    }
  };

  uint64_t sp, dp;
  uint32_t miss; // tier of my row (kTierHost = a cache miss)
  // rows per tier: counted per wave in registers, combined per workgroup in LDS and added to the caller's counters
  // ONCE per workgroup at the end.  (An atomic per wave and tile -- 46 K tiles x 3 tiers on one line -- is served one
  // at a time at the memory side, 12 ns each: 1.6 ms of counter updates behind a 0.5-ms gather, tools/micro_ticket.hip.)
  uint32_t tier_acc[kTierReplica + 1] = {};
  __shared__ unsigned int s_tier[kTierReplica + 1];
  if (miss_count && threadIdx.x <= kTierReplica) s_tier[threadIdx.x] = 0u;
  resolve(wave, sp, dp, miss);
  for (uint64_t tile = wave; tile < num_tiles; tile += num_waves) {
    // software pipeline: the next tile's index -> table -> pointer chain is in flight while this
    // tile's rows stream
    uint64_t sp_n, dp_n;
    uint32_t miss_n;
    resolve(tile + num_waves, sp_n, dp_n, miss_n);

    if (miss_count) {
      tier_acc[0] += (uint32_t)__popcll(__ballot(miss == kTierHost));
      if constexpr (Rows::kTiers) {
#pragma unroll
        for (uint32_t k = kTierRemote; k <= kTierReplica; ++k) tier_acc[k - 1] += (uint32_t)__popcll(__ballot(miss == k));
      }
    }
    const uint64_t row0 = tile * kWave;
    const uint32_t rows_here = (n - row0 < (uint64_t)kWave) ? (uint32_t)(n - row0) : (uint32_t)kWave;
    const uint32_t total = rows_here * rc; // 16-byte chunks in this tile
    const uint64_t out_tile = (uint64_t)(out + row0 * row_bytes);
    for (uint32_t c0 = 0; c0 < total; c0 += kWave * U) {
      V tmp[U];
      uint32_t cc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t c = c0 + u * kWave + lane;
        cc[u] = c < total ? c : total - 1; // clamp: the load is unconditional, the store is not
        // cc / rc, exact for cc * rc < 2^32; rc == 1 has magic = 0 and takes cc itself
        const uint32_t r = __umulhi(cc[u], magic) + (rc == 1 ? cc[u] : 0u);
        const uint32_t col = cc[u] - r * rc;
        const uint64_t p = shfl_u64(sp, (int)r);
        tmp[u] = load_chunk<V, true>(p + (uint64_t)col * CB);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t c = c0 + u * kWave + lane;
        uint64_t q;
        if constexpr (IDENT_DST) {
          q = out_tile + (uint64_t)cc[u] * CB;
        } else {
          const uint32_t r = __umulhi(cc[u], magic) + (rc == 1 ? cc[u] : 0u);
          q = shfl_u64(dp, (int)r) + (uint64_t)(cc[u] - r * rc) * CB;
        }
        if (c < total) store_chunk<V, true>(q, tmp[u]);
      }
    }
    sp = sp_n; dp = dp_n; miss = miss_n;
  }
  if (miss_count) { // uniform: every wave of the workgroup gets here
    __syncthreads(); // s_tier is zeroed
    if (lane == 0) {
#pragma unroll
      for (uint32_t k = 0; k < (Rows::kTiers ? kTierReplica : 1u); ++k)
        if (tier_acc[k]) atomicAdd(&s_tier[k], tier_acc[k]);
    }
    __syncthreads();
    if (threadIdx.x < (Rows::kTiers ? kTierReplica : 1u) && s_tier[threadIdx.x])
      atomicAdd((unsigned long long *)miss_count + threadIdx.x, (unsigned long long)s_tier[threadIdx.x]);
  }
}

// Rows of 8192 chunks or more (>= 128 KiB at 16-byte chunks; the tile sweep's chunk -> row division is exact only
// below that): every row is a long contiguous stream by itself, so one workgroup copies one row.
template <int CB, typename Rows>
__global__ __launch_bounds__(kBlock) void k_gather_long_rows(char *__restrict__ out, Rows rows,
                                                             const uint32_t *__restrict__ dst_index, Count n_arg,
                                                             uint64_t rc, uint64_t *miss_count) {
  using V = typename ChunkT<CB>::type;
  const uint64_t n = n_arg.get();
  for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
    uint32_t tier = 0;
    const uint64_t sp = (uint64_t)rows.row(i, tier);
    const uint64_t dp = (uint64_t)(out + (dst_index ? (uint64_t)dst_index[i] : i) * rc * CB);
    if (miss_count && threadIdx.x == 0) {
      if (tier == kTierHost) atomicAdd((unsigned long long *)miss_count, 1ull);
      else if (Rows::kTiers && tier >= kTierRemote) atomicAdd((unsigned long long *)miss_count + (tier - 1), 1ull);
    }
    for (uint64_t c = threadIdx.x; c < rc; c += kBlock) store_chunk<V, true>(dp + c * CB, load_chunk<V, true>(sp + c * CB));
  }
}

// identity copy of `n` rows (src_index == dst_index == NULL): plain coalesced stream, count on the device
__global__ __launch_bounds__(kBlock) void k_copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src,
                                                       Count n_arg, uint32_t words_per_row) {
  const uint64_t n = n_arg.get() * words_per_row;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) dst[i] = src[i];
}

static inline int pick_chunk(size_t row_bytes, uintptr_t align_bits) {
  for (int cb = 16; cb > 1; cb >>= 1)
    if (row_bytes % cb == 0 && (align_bits % cb) == 0) return cb;
  return 1;
}

} // namespace ggms

// ---- launch timer (include/ggms.h): a pair of events that ride on the next gather's own dispatch packet ------------
struct ggms_launch_timer {
  hipEvent_t start = nullptr, stop = nullptr;
  bool launched = false; // the pair has been attached to a launch (its timestamps mean something)
};

namespace ggms {

static thread_local ggms_launch_timer *tl_armed_timer = nullptr;

// One row-gather launch: with a timer armed on this thread the kernel goes out through hipExtLaunchKernel carrying
// the timer's events (start = the dispatch's own begin timestamp, stop = its completion signal) -- no marker packet
// before or behind it; otherwise the plain launch.
template <typename F, typename... Args>
static inline void launch_rows(F kernel, int grid, hipStream_t stream, Args... args) {
  if (ggms_launch_timer *t = tl_armed_timer) {
    tl_armed_timer = nullptr;
    t->launched = true;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, t->start, t->stop, 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, args...);
  }
}

template <typename Rows>
static int launch_gather(char *out, Rows rows, const uint32_t *dst_index, size_t n_max, Count n,
                         size_t row_bytes, int cb, uint64_t *miss_count, hipStream_t stream) {
  if (n_max == 0) return GGMS_OK;
  const uint64_t rc = row_bytes / cb;
  if (rc == 0) {
    set_error("extract: empty rows");
    return GGMS_ERR_INVALID;
  }
  if (rc >= 8192) {
    const int g = grid_for(n_max, 1);
    switch (cb) {
      case 16: launch_rows(k_gather_long_rows<16, Rows>, g, stream, out, rows, dst_index, n, rc, miss_count); break;
      case 8: launch_rows(k_gather_long_rows<8, Rows>, g, stream, out, rows, dst_index, n, rc, miss_count); break;
      case 4: launch_rows(k_gather_long_rows<4, Rows>, g, stream, out, rows, dst_index, n, rc, miss_count); break;
      case 2: launch_rows(k_gather_long_rows<2, Rows>, g, stream, out, rows, dst_index, n, rc, miss_count); break;
      default: launch_rows(k_gather_long_rows<1, Rows>, g, stream, out, rows, dst_index, n, rc, miss_count); break;
    }
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  const uint32_t magic = rc == 1 ? 0u : (uint32_t)(((1ull << 32) + rc - 1) / rc);
  // one wave per 64 rows, grid-stride.  The grid is capped at ONE 4-wave block per CU (GGMS_EXTRACT_BLOCKS,
  // default 256 -- the one environment variable the library reads: it moves the memory side's share between the
  // gather and the sampler beside it, profiles/r03_ab_gather_grid.txt): with 16 x 16-B loads per lane in flight that
  // already streams at the rate of a full-occupancy launch, and the free wave slots let the next batch's
  // latency-bound sampling kernels run beside the gather on another stream.
  static const int max_blocks = [] { const char *e = getenv("GGMS_EXTRACT_BLOCKS"); int v = e ? atoi(e) : 256; return v > 0 ? v : 256; }();
  int grid = grid_for(n_max, kBlock);
  if (grid > max_blocks) grid = max_blocks;
  // 16 independent chunk loads per lane once a row has >= 8 chunks, else 8 (measured: 0.58 -> 0.60 of peak at 400-B
  // rows, 0.65 -> 0.70 at 512-B rows, and the gather holds its rate when the sampler runs beside it)
  const bool deep = rc >= 8;
#define GGMS_LAUNCH(CB, ID)                                                                                          \
  do {                                                                                                               \
    if (deep)                                                                                                        \
      launch_rows(k_gather_rows<CB, Rows, ID, 16>, grid, stream, out, rows, dst_index, n, (uint32_t)rc, magic, miss_count); \
    else                                                                                                             \
      launch_rows(k_gather_rows<CB, Rows, ID, 8>, grid, stream, out, rows, dst_index, n, (uint32_t)rc, magic, miss_count); \
  } while (0)
#define GGMS_CASE(CB)                                        \
  case CB:                                                   \
    if (dst_index == nullptr) GGMS_LAUNCH(CB, true);         \
    else GGMS_LAUNCH(CB, false);                             \
    break;
  switch (cb) {
    GGMS_CASE(16)
    GGMS_CASE(8)
    GGMS_CASE(4)
    GGMS_CASE(2)
    GGMS_CASE(1)
  }
#undef GGMS_CASE
#undef GGMS_LAUNCH
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // namespace ggms

using namespace ggms;

extern "C" {

int ggms_extract(void *dst, const void *src, const ggms_id_t *index, size_t num_index, size_t dim, int dtype,
                 ggms_stream_t stream) {
  const size_t es = ggms_dtype_bytes(dtype);
  GGMS_CHECK_ARG(es != 0 && dim != 0);
  if (num_index == 0) return GGMS_OK;
  GGMS_CHECK_ARG(dst && src && index);
  const size_t row_bytes = dim * es;
  const int cb = pick_chunk(row_bytes, (uintptr_t)dst | (uintptr_t)src);
  PlainRows rows{(const char *)src, index, row_bytes};
  return launch_gather((char *)dst, rows, nullptr, num_index, count_of(num_index), row_bytes, cb, nullptr,
                       to_stream(stream));
}

int ggms_gather_scatter(void *out, const void *src, const ggms_id_t *src_index, const ggms_id_t *dst_index,
                        size_t num, const uint64_t *num_dev, size_t dim, int dtype, ggms_stream_t stream) {
  const size_t es = ggms_dtype_bytes(dtype);
  GGMS_CHECK_ARG(es != 0 && dim != 0);
  if (num == 0) return GGMS_OK;
  GGMS_CHECK_ARG(out && src);
  const size_t row_bytes = dim * es;
  if (!src_index && !dst_index && row_bytes % 4 == 0 && (((uintptr_t)out | (uintptr_t)src) & 3) == 0) {
    launch_rows(k_copy_words, grid_for(num * (row_bytes / 4), kBlock * 4), to_stream(stream), (uint32_t *)out,
                (const uint32_t *)src, count_of(num, num_dev), (uint32_t)(row_bytes / 4));
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  const int cb = pick_chunk(row_bytes, (uintptr_t)out | (uintptr_t)src);
  PlainRows rows{(const char *)src, src_index, row_bytes};
  return launch_gather((char *)out, rows, dst_index, num, count_of(num, num_dev), row_bytes, cb, nullptr,
                       to_stream(stream));
}

// ggms_gather_scatter with the source row taken as src_index[i] & src_row_mask: gpu_mock_extract
// (cuda_extraction.cu:51-70) and the mock-table forms of the miss extract (cuda_cache_manager_host.cc:47-48)
int ggms_gather_scatter_masked(void *out, const void *src, const ggms_id_t *src_index, const ggms_id_t *dst_index,
                               size_t num, const uint64_t *num_dev, size_t dim, int dtype, uint32_t src_row_mask,
                               ggms_stream_t stream) {
  const size_t es = ggms_dtype_bytes(dtype);
  GGMS_CHECK_ARG(es != 0 && dim != 0);
  if (num == 0) return GGMS_OK;
  GGMS_CHECK_ARG(out && src && src_index);
  const size_t row_bytes = dim * es;
  const int cb = pick_chunk(row_bytes, (uintptr_t)out | (uintptr_t)src);
  PlainRows rows{(const char *)src, src_index, row_bytes, src_row_mask};
  return launch_gather((char *)out, rows, dst_index, num, count_of(num, num_dev), row_bytes, cb, nullptr,
                       to_stream(stream));
}

// GPUMockExtract (cuda_extraction.cu:119-160): dst[i, :] = src[index[i] & (2^mock_bits - 1), :]
int ggms_mock_extract(void *dst, const void *src, const ggms_id_t *index, size_t num_index, size_t dim, int dtype,
                      uint32_t mock_bits, ggms_stream_t stream) {
  GGMS_CHECK_ARG(mock_bits >= 1 && mock_bits <= 32);
  const uint32_t mask = mock_bits == 32 ? 0xffffffffu : ((1u << mock_bits) - 1u);
  return ggms_gather_scatter_masked(dst, src, index, nullptr, num_index, nullptr, dim, dtype, mask, stream);
}

int ggms_gather_scatter_partition(void *out, const void *const *parts, uint32_t num_part,
                                  const ggms_id_t *src_index, const ggms_id_t *dst_index, size_t num,
                                  const uint64_t *num_dev, size_t dim, int dtype, ggms_stream_t stream) {
  const size_t es = ggms_dtype_bytes(dtype);
  GGMS_CHECK_ARG(es != 0 && dim != 0 && num_part != 0);
  if (num == 0) return GGMS_OK;
  GGMS_CHECK_ARG(out && parts && src_index);
  const size_t row_bytes = dim * es;
  // shard bases come from hipMalloc / hipIpcOpenMemHandle / hipHostMalloc: >= 256-B aligned
  const int cb = pick_chunk(row_bytes, (uintptr_t)out);
  PartPtrs pp;
  if (!part_ptrs(parts, num_part, pp)) return GGMS_ERR_INVALID;
  PartitionRows rows{pp, src_index, row_bytes, divisor_of(num_part)};
  return launch_gather((char *)out, rows, dst_index, num, count_of(num, num_dev), row_bytes, cb, nullptr,
                       to_stream(stream));
}

int ggms_extract_cached(void *out, const ggms_id_t *nodes, size_t num_nodes, const uint64_t *num_nodes_dev,
                        const ggms_id_t *table, const void *const *parts, uint32_t num_part,
                        const void *host_feat, size_t dim, int dtype, uint64_t *num_miss_dev,
                        ggms_stream_t stream) {
  const size_t es = ggms_dtype_bytes(dtype);
  GGMS_CHECK_ARG(es != 0 && dim != 0);
  if (num_nodes == 0) return GGMS_OK;
  GGMS_CHECK_ARG(out && nodes && parts);
  const size_t row_bytes = dim * es;
  PartPtrs pp;
  if (!part_ptrs(parts, num_part, pp)) return GGMS_ERR_INVALID;
  const Divisor div = divisor_of(num_part ? num_part : 1);
  if (!table) { // full cache in node order: slot = node id
    IdentRows rows{pp, nodes, row_bytes, div};
    if (num_miss_dev) GGMS_HIP(hipMemsetAsync(num_miss_dev, 0, sizeof(uint64_t), to_stream(stream)));
    return launch_gather((char *)out, rows, nullptr, num_nodes, count_of(num_nodes, num_nodes_dev), row_bytes,
                         pick_chunk(row_bytes, (uintptr_t)out), nullptr, to_stream(stream));
  }
  const int cb = pick_chunk(row_bytes, (uintptr_t)out | (uintptr_t)host_feat);
  CachedRows rows{pp, nodes, table, (const char *)host_feat, row_bytes, div};
  return launch_gather((char *)out, rows, nullptr, num_nodes, count_of(num_nodes, num_nodes_dev), row_bytes,
                       cb, num_miss_dev, to_stream(stream));
}

int ggms_extract_tiered(void *out, const ggms_id_t *nodes, size_t num_nodes, const uint64_t *num_nodes_dev,
                        const ggms_feature_tiers_t *tiers, size_t dim, int dtype, uint64_t *tier_rows_dev,
                        ggms_stream_t stream) {
  const size_t es = ggms_dtype_bytes(dtype);
  GGMS_CHECK_ARG(es != 0 && dim != 0 && tiers);
  if (num_nodes == 0) return GGMS_OK;
  GGMS_CHECK_ARG(out && nodes && tiers->parts && tiers->num_part >= 1 && tiers->my_part < tiers->num_part);
  GGMS_CHECK_ARG(tiers->num_replica == 0 || tiers->replica);
  GGMS_CHECK_ARG(tiers->num_replica < (1ull << 32));
  const size_t row_bytes = dim * es;
  const int cb = pick_chunk(row_bytes, (uintptr_t)out | (uintptr_t)tiers->host_feat | (uintptr_t)tiers->replica);
  PartPtrs pp;
  if (!part_ptrs(tiers->parts, tiers->num_part, pp)) return GGMS_ERR_INVALID;
  TieredRows rows{pp, nodes, tiers->table, (const char *)tiers->host_feat,
                  (const char *)tiers->replica, row_bytes, (uint32_t)tiers->num_replica, divisor_of(tiers->num_part),
                  tiers->my_part, tiers->host_row_mask ? tiers->host_row_mask : 0xffffffffu};
  return launch_gather((char *)out, rows, nullptr, num_nodes, count_of(num_nodes, num_nodes_dev), row_bytes, cb,
                       tier_rows_dev, to_stream(stream));
}

// ---- launch timer ---------------------------------------------------------------------------------------------------
int ggms_launch_timer_create(ggms_launch_timer_t **timer) {
  GGMS_CHECK_ARG(timer);
  ggms_launch_timer *t = new ggms_launch_timer();
  hipError_t e = hipEventCreate(&t->start);
  if (e == hipSuccess) e = hipEventCreate(&t->stop);
  if (e != hipSuccess) {
    if (t->start) (void)hipEventDestroy(t->start);
    delete t;
    set_error("ggms_launch_timer_create: hipEventCreate -> %s", hipGetErrorString(e));
    return GGMS_ERR_HIP;
  }
  *timer = t;
  return GGMS_OK;
}

int ggms_launch_timer_destroy(ggms_launch_timer_t *timer) {
  if (!timer) return GGMS_OK;
  if (tl_armed_timer == timer) tl_armed_timer = nullptr;
  const hipError_t e0 = hipEventDestroy(timer->start), e1 = hipEventDestroy(timer->stop);
  delete timer;
  GGMS_HIP(e0);
  GGMS_HIP(e1);
  return GGMS_OK;
}

int ggms_launch_timer_arm(ggms_launch_timer_t *timer) {
  GGMS_CHECK_ARG(timer);
  timer->launched = false;
  tl_armed_timer = timer;
  return GGMS_OK;
}

int ggms_launch_timer_wait(ggms_launch_timer_t *timer, ggms_stream_t stream) {
  GGMS_CHECK_ARG(timer);
  if (timer->launched) GGMS_HIP(hipStreamWaitEvent(to_stream(stream), timer->stop, 0));
  return GGMS_OK;
}

int ggms_launch_timer_elapsed_us(ggms_launch_timer_t *timer, double *us) {
  GGMS_CHECK_ARG(timer && us);
  if (!timer->launched) {
    set_error("ggms_launch_timer_elapsed_us: the timer never rode a launch (armed on another thread, or the call had no rows)");
    return GGMS_ERR_INVALID;
  }
  GGMS_HIP(hipEventSynchronize(timer->stop));
  float ms = 0.f;
  GGMS_HIP(hipEventElapsedTime(&ms, timer->start, timer->stop));
  *us = (double)ms * 1e3;
  return GGMS_OK;
}

int ggms_launch_timer_span_us(ggms_launch_timer_t *first, ggms_launch_timer_t *last, double *us) {
  GGMS_CHECK_ARG(first && last && us);
  if (!first->launched || !last->launched) {
    set_error("ggms_launch_timer_span_us: a timer that never rode a launch");
    return GGMS_ERR_INVALID;
  }
  GGMS_HIP(hipEventSynchronize(first->stop));
  GGMS_HIP(hipEventSynchronize(last->stop));
  float ms = 0.f;
  GGMS_HIP(hipEventElapsedTime(&ms, first->start, last->stop));
  *us = (double)ms * 1e3;
  return GGMS_OK;
}

} // extern "C"
