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
//   * U independent 16-B loads per lane are issued before the first store.
// Algorithmic bytes per row: 4 (index) + 2 * row_bytes.
#include "ggms_device.h"

namespace ggms {

template <int BYTES> struct ChunkT;
template <> struct ChunkT<16> { using type = uint4; };
template <> struct ChunkT<8> { using type = uint2; };
template <> struct ChunkT<4> { using type = uint32_t; };
template <> struct ChunkT<2> { using type = uint16_t; };
template <> struct ChunkT<1> { using type = uint8_t; };

// ---- row locators -----------------------------------------------------------
// src(i) = index ? index[i] : i ;  row pointer = base + src(i) * row_bytes
struct PlainRows {
  const char *base;
  const uint32_t *index;
  uint64_t row_bytes;
  __device__ __forceinline__ const char *row(uint64_t i, bool &miss) const {
    miss = false;
    const uint64_t s = index ? (uint64_t)index[i] : i;
    return base + s * row_bytes;
  }
};

// DeviceDistFeature::Get (cuda/dist_graph.h:191-205): slot -> shard slot % P, row slot / P
struct PartitionRows {
  const char *const *parts;
  const uint32_t *index;
  uint64_t row_bytes;
  uint32_t num_part;
  __device__ __forceinline__ const char *row(uint64_t i, bool &miss) const {
    miss = false;
    const uint32_t slot = index[i];
    const uint32_t part = slot % num_part, real = slot / num_part;
    return parts[part] + (uint64_t)real * row_bytes;
  }
};

// fused hit/miss: table[node] == kEmptyKey -> host tier row `node`, else cache slot
struct CachedRows {
  const char *const *parts;
  const uint32_t *nodes;
  const uint32_t *table;
  const char *host;
  uint64_t row_bytes;
  uint32_t num_part; // 0: one cache array parts[0]
  __device__ __forceinline__ const char *row(uint64_t i, bool &miss) const {
    const uint32_t node = nodes[i];
    const uint32_t slot = table[node];
    miss = (slot == kEmptyKey);
    if (miss) return host + (uint64_t)node * row_bytes;
    if (num_part == 0) return parts[0] + (uint64_t)slot * row_bytes;
    const uint32_t part = slot % num_part, real = slot / num_part;
    return parts[part] + (uint64_t)real * row_bytes;
  }
};

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  const uint32_t lo = __shfl((uint32_t)v, src, 64);
  const uint32_t hi = __shfl((uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}

template <int CB, typename Rows>
__global__ __launch_bounds__(kBlock) void k_gather_rows(char *__restrict__ out, Rows rows,
                                                        const uint32_t *__restrict__ dst_index, Count n_arg,
                                                        uint32_t rc, uint64_t magic, uint64_t *miss_count) {
  using V = typename ChunkT<CB>::type;
  constexpr int U = (CB >= 8) ? 4 : 8;
  const uint64_t n = n_arg.get();
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const uint64_t num_waves = (uint64_t)gridDim.x * (kBlock / kWave);
  const uint64_t row_bytes = (uint64_t)rc * CB;

  for (uint64_t tile = wave; tile * kWave < n; tile += num_waves) {
    const uint64_t row0 = tile * kWave;
    const uint64_t my_row = row0 + lane;
    uint64_t sp = 0, dp = 0;
    bool miss = false;
    if (my_row < n) {
      sp = (uint64_t)rows.row(my_row, miss);
      const uint64_t drow = dst_index ? (uint64_t)dst_index[my_row] : my_row;
      dp = (uint64_t)(out + drow * row_bytes);
    }
    if (miss_count) {
      const uint64_t m = __ballot(miss);
      if (lane == 0 && m) atomicAdd((unsigned long long *)miss_count, (unsigned long long)__popcll(m));
    }
    const uint32_t rows_here = (n - row0 < (uint64_t)kWave) ? (uint32_t)(n - row0) : (uint32_t)kWave;
    const uint32_t total = rows_here * rc;
    for (uint32_t c0 = 0; c0 < total; c0 += kWave * U) {
      V tmp[U];
      uint32_t col[U];
      uint32_t own[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t c = c0 + u * kWave + lane;
        uint32_t r = (uint32_t)(((uint64_t)c * magic) >> 32); // c / rc, exact for c * rc < 2^32
        r = r < (uint32_t)kWave ? r : (uint32_t)(kWave - 1);
        own[u] = r;
        col[u] = c - r * rc;
        const uint64_t p = shfl_u64(sp, (int)r);
        if (c < total) tmp[u] = reinterpret_cast<const V *>(p)[col[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t c = c0 + u * kWave + lane;
        const uint64_t d = shfl_u64(dp, (int)own[u]);
        if (c < total) reinterpret_cast<V *>(d)[col[u]] = tmp[u];
      }
    }
  }
}

static inline int pick_chunk(size_t row_bytes, uintptr_t align_bits) {
  for (int cb = 16; cb > 1; cb >>= 1)
    if (row_bytes % cb == 0 && (align_bits % cb) == 0) return cb;
  return 1;
}

template <typename Rows>
static int launch_gather(char *out, Rows rows, const uint32_t *dst_index, size_t n_max, Count n,
                         size_t row_bytes, int cb, uint64_t *miss_count, hipStream_t stream) {
  if (n_max == 0) return GGMS_OK;
  const uint64_t rc = row_bytes / cb;
  if (rc == 0 || rc >= 8192) {
    set_error("extract: row of %zu bytes in %d-byte chunks is outside the supported range", row_bytes, cb);
    return GGMS_ERR_INVALID;
  }
  const uint64_t magic = ((1ull << 32) + rc - 1) / rc;
  const int grid = grid_for(n_max, kBlock); // one wave per 64 rows
  switch (cb) {
#define GGMS_CASE(CB)                                                                              \
  case CB:                                                                                         \
    hipLaunchKernelGGL((k_gather_rows<CB, Rows>), dim3(grid), dim3(kBlock), 0, stream, out, rows,  \
                       dst_index, n, (uint32_t)rc, magic, miss_count);                             \
    break;
    GGMS_CASE(16)
    GGMS_CASE(8)
    GGMS_CASE(4)
    GGMS_CASE(2)
    GGMS_CASE(1)
#undef GGMS_CASE
  }
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
  const int cb = pick_chunk(row_bytes, (uintptr_t)out | (uintptr_t)src);
  PlainRows rows{(const char *)src, src_index, row_bytes};
  return launch_gather((char *)out, rows, dst_index, num, count_of(num, num_dev), row_bytes, cb, nullptr,
                       to_stream(stream));
}

int ggms_gather_scatter_partition(void *out, const void *const *parts_dev, uint32_t num_part,
                                  const ggms_id_t *src_index, const ggms_id_t *dst_index, size_t num,
                                  const uint64_t *num_dev, size_t dim, int dtype, ggms_stream_t stream) {
  const size_t es = ggms_dtype_bytes(dtype);
  GGMS_CHECK_ARG(es != 0 && dim != 0 && num_part != 0);
  if (num == 0) return GGMS_OK;
  GGMS_CHECK_ARG(out && parts_dev && src_index);
  const size_t row_bytes = dim * es;
  // shard bases come from hipMalloc / hipIpcOpenMemHandle / hipHostMalloc: >= 256-B aligned
  const int cb = pick_chunk(row_bytes, (uintptr_t)out);
  PartitionRows rows{(const char *const *)parts_dev, src_index, row_bytes, num_part};
  return launch_gather((char *)out, rows, dst_index, num, count_of(num, num_dev), row_bytes, cb, nullptr,
                       to_stream(stream));
}

int ggms_extract_cached(void *out, const ggms_id_t *nodes, size_t num_nodes, const uint64_t *num_nodes_dev,
                        const ggms_id_t *table, const void *const *parts_dev, uint32_t num_part,
                        const void *host_feat, size_t dim, int dtype, uint64_t *num_miss_dev,
                        ggms_stream_t stream) {
  const size_t es = ggms_dtype_bytes(dtype);
  GGMS_CHECK_ARG(es != 0 && dim != 0);
  if (num_nodes == 0) return GGMS_OK;
  GGMS_CHECK_ARG(out && nodes && table && parts_dev);
  const size_t row_bytes = dim * es;
  const int cb = pick_chunk(row_bytes, (uintptr_t)out | (uintptr_t)host_feat);
  CachedRows rows{(const char *const *)parts_dev, nodes, table, (const char *)host_feat, row_bytes, num_part};
  return launch_gather((char *)out, rows, nullptr, num_nodes, count_of(num_nodes, num_nodes_dev), row_bytes,
                       cb, num_miss_dev, to_stream(stream));
}

} // extern "C"
