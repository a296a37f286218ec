// cache.hip -- feature-cache hit/miss split.
//
// Reference: GetMissCacheIndex (cuda/cuda_cache_manager_device.cu:355-441) =
// count_miss_cache :40-75 + 2 x cub::DeviceScan + get_miss_index :77-117 +
// get_cache_index :119-169, with two D2H copies of the totals.
// Here: ONE stable scan of the miss flags.  A hit's position in the hit list is
// i - (number of misses before i), so both lists come out of the same pass, in
// input order, and both totals stay on the device.
#include <algorithm>
#include <cstring>

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

// ---- exchange form of the remote gather: split a batch by owning shard ----------------------------
__device__ __forceinline__ uint32_t owner_of(uint32_t slot, uint32_t num_part) {
  return slot == kEmptyKey ? num_part : slot % num_part;
}

// Counters shared by the whole grid are touched ONCE per workgroup and bucket: atomics on one address are served one
// at a time at the memory side (12 ns each, tools/micro_ticket.hip) -- one per wave and bucket was 400 K of them per
// papers100M batch on nine neighbouring words.  Waves count into LDS; a workgroup owns a contiguous piece of the
// batch (kOwnerGrid pieces), so that the bucket kernel can claim its slice of every bucket with one returning atomic
// and then place its rows with LDS cursors only.
constexpr int kOwnerGrid = 256;
constexpr uint32_t kOwnerBuckets = 65; // num_part <= 64, + the host tier

__device__ __forceinline__ void owner_piece(uint64_t n, uint64_t &lo, uint64_t &hi) { // this workgroup's items
  const uint64_t per = ((n + gridDim.x - 1) / gridDim.x + kBlock - 1) / kBlock * kBlock; // whole block rounds
  lo = (uint64_t)blockIdx.x * per;
  hi = lo + per < n ? lo + per : n;
}

__global__ __launch_bounds__(kBlock) void k_owner_histogram(const uint32_t *__restrict__ table,
                                                            const uint32_t *__restrict__ nodes, Count n_arg,
                                                            uint32_t num_part, uint32_t *__restrict__ slots_out,
                                                            unsigned long long *counts) {
  __shared__ unsigned int s_cnt[kOwnerBuckets];
  const uint64_t n = n_arg.get();
  for (uint32_t p = threadIdx.x; p <= num_part; p += kBlock) s_cnt[p] = 0u;
  __syncthreads();
  uint64_t lo, hi;
  owner_piece(n, lo, hi);
  for (uint64_t i0 = lo; i0 < hi; i0 += kBlock) { // whole waves stay together for the ballots
    const uint64_t i = i0 + threadIdx.x;
    uint32_t owner = 0xffffffffu;
    if (i < hi) {
      const uint32_t slot = table ? table[nodes[i]] : nodes[i]; // no table: every node cached at slot = node id
      slots_out[i] = slot;
      owner = owner_of(slot, num_part);
    }
    for (uint32_t p = 0; p <= num_part; ++p) {
      const uint64_t m = __ballot(owner == p);
      if (m && lane_id() == (uint32_t)__builtin_ctzll(m)) atomicAdd(&s_cnt[p], (unsigned int)__popcll(m));
    }
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p <= num_part; p += kBlock)
    if (s_cnt[p]) atomicAdd(&counts[p], (unsigned long long)s_cnt[p]);
}

__global__ __launch_bounds__(kBlock) void k_owner_bucket(const uint32_t *__restrict__ slots,
                                                         const uint32_t *__restrict__ nodes, Count n_arg,
                                                         uint32_t num_part, unsigned long long *cursor,
                                                         uint32_t *__restrict__ bucket_row,
                                                         uint32_t *__restrict__ bucket_pos) {
  __shared__ unsigned int s_cnt[kOwnerBuckets];
  __shared__ unsigned long long s_base[kOwnerBuckets];
  const uint64_t n = n_arg.get();
  for (uint32_t p = threadIdx.x; p <= num_part; p += kBlock) s_cnt[p] = 0u;
  __syncthreads();
  uint64_t lo, hi;
  owner_piece(n, lo, hi);
  // pass 1: this piece's rows per bucket
  for (uint64_t i0 = lo; i0 < hi; i0 += kBlock) {
    const uint64_t i = i0 + threadIdx.x;
    const uint32_t owner = i < hi ? owner_of(slots[i], num_part) : 0xffffffffu;
    for (uint32_t p = 0; p <= num_part; ++p) {
      const uint64_t m = __ballot(owner == p);
      if (m && lane_id() == (uint32_t)__builtin_ctzll(m)) atomicAdd(&s_cnt[p], (unsigned int)__popcll(m));
    }
  }
  __syncthreads();
  // claim a slice of every bucket (one returning atomic per bucket), then hand it out from LDS
  for (uint32_t p = threadIdx.x; p <= num_part; p += kBlock) {
    s_base[p] = s_cnt[p] ? atomicAdd(&cursor[p], (unsigned long long)s_cnt[p]) : 0ull;
    s_cnt[p] = 0u;
  }
  __syncthreads();
  for (uint64_t i0 = lo; i0 < hi; i0 += kBlock) {
    const uint64_t i = i0 + threadIdx.x;
    uint32_t owner = 0xffffffffu, slot = 0;
    if (i < hi) {
      slot = slots[i];
      owner = owner_of(slot, num_part);
    }
    for (uint32_t p = 0; p <= num_part; ++p) {
      const uint64_t m = __ballot(owner == p);
      if (!m) continue;
      const uint32_t leader = (uint32_t)__builtin_ctzll(m);
      unsigned int off = 0;
      if (lane_id() == leader) off = atomicAdd(&s_cnt[p], (unsigned int)__popcll(m));
      off = __shfl(off, (int)leader, 64);
      if (owner == p) {
        const uint64_t at = s_base[p] + off + __popcll(m & ((1ull << lane_id()) - 1ull));
        bucket_row[at] = (p == num_part) ? nodes[i] : slot / num_part;
        bucket_pos[at] = (uint32_t)i;
      }
    }
  }
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

int ggms_get_miss_cache_index_dev(const ggms_id_t *table, const ggms_id_t *nodes, size_t num_nodes,
                                  const uint64_t *num_nodes_dev, ggms_id_t *miss_src_index, ggms_id_t *miss_dst_index,
                                  uint64_t *num_miss_dev, ggms_id_t *cache_src_index, ggms_id_t *cache_dst_index,
                                  uint64_t *num_cache_dev, void *workspace, size_t workspace_bytes,
                                  ggms_stream_t stream) {
  GGMS_CHECK_ARG(num_miss_dev && num_cache_dev);
  hipStream_t s = to_stream(stream);
  GGMS_HIP(hipMemsetAsync(num_miss_dev, 0, sizeof(uint64_t), s));
  GGMS_HIP(hipMemsetAsync(num_cache_dev, 0, sizeof(uint64_t), s));
  if (num_nodes == 0) return GGMS_OK;
  GGMS_CHECK_ARG(table && nodes && miss_src_index && miss_dst_index && cache_src_index && cache_dst_index);
  GGMS_CHECK_ARG(workspace && workspace_bytes >= ggms_cache_index_workspace_bytes(num_nodes));
  GGMS_CHECK_ARG(num_nodes < (1ull << 32));
  const Count n = count_of(num_nodes, num_nodes_dev);
  return tile_scan(MissFlag{table, nodes},
                   SplitEmit{table, nodes, miss_src_index, miss_dst_index, cache_src_index, cache_dst_index,
                             num_miss_dev, num_cache_dev, n},
                   num_nodes, n, ScanArea{(uint32_t *)workspace, false}, nullptr, nullptr, nullptr, s);
}

int ggms_get_miss_cache_index(const ggms_id_t *table, const ggms_id_t *nodes, size_t num_nodes,
                              ggms_id_t *miss_src_index, ggms_id_t *miss_dst_index, uint64_t *num_miss_dev,
                              ggms_id_t *cache_src_index, ggms_id_t *cache_dst_index, uint64_t *num_cache_dev,
                              void *workspace, size_t workspace_bytes, ggms_stream_t stream) {
  return ggms_get_miss_cache_index_dev(table, nodes, num_nodes, nullptr, miss_src_index, miss_dst_index, num_miss_dev,
                                       cache_src_index, cache_dst_index, num_cache_dev, workspace, workspace_bytes, stream);
}

int ggms_owner_histogram(const ggms_id_t *table, const ggms_id_t *nodes, size_t num_nodes,
                         const uint64_t *num_nodes_dev, uint32_t num_part, ggms_id_t *slots_out, uint64_t *counts_dev,
                         ggms_stream_t stream) {
  GGMS_CHECK_ARG(num_part >= 1 && num_part <= 64 && counts_dev);
  if (num_nodes == 0) return GGMS_OK;
  GGMS_CHECK_ARG(nodes && slots_out && num_nodes < (1ull << 32)); // table == NULL: identity slots (full cache)
  hipLaunchKernelGGL(k_owner_histogram, dim3(std::min(grid_for(num_nodes, kBlock), kOwnerGrid)), dim3(kBlock), 0, to_stream(stream), table,
                     nodes, count_of(num_nodes, num_nodes_dev), num_part, slots_out, (unsigned long long *)counts_dev);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

int ggms_owner_bucket(const ggms_id_t *slots, const ggms_id_t *nodes, size_t num_nodes, const uint64_t *num_nodes_dev,
                      uint32_t num_part, uint64_t *cursor_dev, ggms_id_t *bucket_row, ggms_id_t *bucket_pos,
                      ggms_stream_t stream) {
  GGMS_CHECK_ARG(num_part >= 1 && num_part <= 64 && cursor_dev);
  if (num_nodes == 0) return GGMS_OK;
  GGMS_CHECK_ARG(slots && nodes && bucket_row && bucket_pos && num_nodes < (1ull << 32));
  hipLaunchKernelGGL(k_owner_bucket, dim3(std::min(grid_for(num_nodes, kBlock), kOwnerGrid)), dim3(kBlock), 0, to_stream(stream), slots, nodes,
                     count_of(num_nodes, num_nodes_dev), num_part, (unsigned long long *)cursor_dev, bucket_row,
                     bucket_pos);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

// ---- shards across processes: cuda/dist_graph.cu:228-272 ----
size_t ggms_ipc_safe_bytes(size_t bytes) {
  return (bytes & 0x80000000ull) ? ((bytes | 0xffffffffull) + 1ull) : bytes;
}

int ggms_device_alloc(void **ptr, size_t bytes) {
  GGMS_CHECK_ARG(ptr && bytes > 0);
  GGMS_HIP(hipMalloc(ptr, ggms_ipc_safe_bytes(bytes))); // a size the peers can open (include/ggms.h)
  return GGMS_OK;
}

int ggms_device_free(void *ptr) {
  if (ptr) GGMS_HIP(hipFree(ptr));
  return GGMS_OK;
}

int ggms_ipc_export(const void *ptr, void *handle) {
  GGMS_CHECK_ARG(ptr && handle);
  static_assert(sizeof(hipIpcMemHandle_t) <= GGMS_IPC_HANDLE_BYTES, "handle buffer too small");
  hipIpcMemHandle_t h;
  GGMS_HIP(hipIpcGetMemHandle(&h, const_cast<void *>(ptr)));
  memset(handle, 0, GGMS_IPC_HANDLE_BYTES);
  memcpy(handle, &h, sizeof(h));
  return GGMS_OK;
}

int ggms_ipc_import(const void *handle, void **ptr) {
  GGMS_CHECK_ARG(ptr && handle);
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  GGMS_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
  return GGMS_OK;
}

int ggms_ipc_release(void *ptr) {
  if (ptr) GGMS_HIP(hipIpcCloseMemHandle(ptr));
  return GGMS_OK;
}

} // extern "C"
