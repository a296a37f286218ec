// cache.hip -- feature-cache hit/miss split.
//
// Reference: GetMissCacheIndex (cuda/cuda_cache_manager_device.cu:355-441) =
// count_miss_cache :40-75 + 2 x cub::DeviceScan + get_miss_index :77-117 +
// get_cache_index :119-169, with two D2H copies of the totals.
// Here: ONE stable scan of the miss flags.  A hit's position in the hit list is
// i - (number of misses before i), so both lists come out of the same pass, in
// input order, and both totals stay on the device.
#include "tile_scan.h"

namespace ggms {

struct MissFlag {
  const uint32_t *table;
  const uint32_t *nodes;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
    return table[nodes[i]] == kEmptyKey ? 1u : 0u;
  }
};

struct SplitEmit {
  const uint32_t *table;
  const uint32_t *nodes;
  uint32_t *miss_src, *miss_dst, *hit_src, *hit_dst;
  uint64_t *num_miss, *num_hit;
  Count n;
  __device__ __forceinline__ void operator()(uint64_t i, uint32_t miss, uint32_t misses_before) const {
    const uint32_t node = nodes[i];
    if (miss) {
      miss_dst[misses_before] = (uint32_t)i; // row in the batch output
      miss_src[misses_before] = node;        // row in the full (host) table
    } else {
      const uint32_t h = (uint32_t)i - misses_before;
      hit_dst[h] = (uint32_t)i;
      hit_src[h] = table[node];              // cache slot
    }
    if (i + 1 == n.get()) {
      *num_miss = misses_before + miss;
      *num_hit = (i + 1) - (misses_before + miss);
    }
  }
};

// presample: freq[node] += 1 for every node of the batch (dist/pre_sampler.cc:101-105 does this on the host)
__global__ __launch_bounds__(kBlock) void k_count_nodes(uint32_t *__restrict__ freq, const uint32_t *__restrict__ nodes,
                                                        Count n_arg) {
  const uint64_t n = n_arg.get();
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    atomicAdd(&freq[nodes[i]], 1u);
}

} // namespace ggms

using namespace ggms;

extern "C" {

int ggms_count_nodes(uint32_t *freq, const ggms_id_t *nodes, size_t num_nodes, const uint64_t *num_nodes_dev,
                     ggms_stream_t stream) {
  if (num_nodes == 0) return GGMS_OK;
  GGMS_CHECK_ARG(freq && nodes);
  hipLaunchKernelGGL(k_count_nodes, dim3(grid_for(num_nodes, kBlock)), dim3(kBlock), 0, to_stream(stream), freq, nodes,
                     count_of(num_nodes, num_nodes_dev));
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

size_t ggms_cache_index_workspace_bytes(size_t num_nodes) {
  return (tile_scan_words(num_nodes) + 16) * sizeof(uint32_t);
}

int ggms_get_miss_cache_index(const ggms_id_t *table, const ggms_id_t *nodes, size_t num_nodes,
                              ggms_id_t *miss_src_index, ggms_id_t *miss_dst_index, uint64_t *num_miss_dev,
                              ggms_id_t *cache_src_index, ggms_id_t *cache_dst_index, uint64_t *num_cache_dev,
                              void *workspace, size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(num_miss_dev && num_cache_dev);
  hipStream_t s = to_stream(stream);
  GGMS_HIP(hipMemsetAsync(num_miss_dev, 0, sizeof(uint64_t), s));
  GGMS_HIP(hipMemsetAsync(num_cache_dev, 0, sizeof(uint64_t), s));
  if (num_nodes == 0) return GGMS_OK;
  GGMS_CHECK_ARG(table && nodes && miss_src_index && miss_dst_index && cache_src_index && cache_dst_index);
  GGMS_CHECK_ARG(workspace && workspace_bytes >= ggms_cache_index_workspace_bytes(num_nodes));
  GGMS_CHECK_ARG(num_nodes < (1ull << 32));
  const Count n = count_of(num_nodes);
  return tile_scan(MissFlag{table, nodes},
                   SplitEmit{table, nodes, miss_src_index, miss_dst_index, cache_src_index, cache_dst_index,
                             num_miss_dev, num_cache_dev, n},
                   num_nodes, n, ScanArea{(uint32_t *)workspace, false}, nullptr, nullptr, nullptr, s);
}

} // extern "C"
