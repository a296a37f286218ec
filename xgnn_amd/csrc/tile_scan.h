// tile_scan.h -- ordered (stable) device-wide exclusive scan / compaction frame.
//
// Replaces the reference's {count kernel, cub::DeviceScan::ExclusiveSum over
// grid+1 entries, compact kernel with cub::BlockScan} triple
// (e.g. cuda_sampling_khop3.cu:148-230,286-295; cuda_hashtable.cu:197-232,
// 406-458; cuda_cache_manager_device.cu:40-169) with three launches that never
// return to the host: the element count may live in device memory.
//
//   phase 1  k_tile_reduce : tile_sums[t]   = sum of value(i) over tile t
//   phase 2  k_tile_prefix : tile_prefix[t] = base + exclusive sum (one block)
//   phase 3  k_tile_apply  : emit(i, value(i), global exclusive prefix of i)
//
// Tile = 1024 items = 4 rounds of 256 threads (coalesced, item = tile*1024 +
// round*256 + thread), the same item->position mapping as the reference's
// kCudaTileSize/kCudaBlockSize idiom, so output order is the input order.
#pragma once

#include "ggms_device.h"

namespace ggms {

constexpr uint32_t kTile = 1024;

inline size_t num_tiles_for(size_t n) { return (n + kTile - 1) / kTile; }
// scratch for tile_sums + tile_prefix (+1 each), in uint32 words
inline size_t tile_scan_words(size_t n) { return 2 * (num_tiles_for(n) + 2); }

template <typename ValueF>
__global__ __launch_bounds__(kBlock) void k_tile_reduce(ValueF value, Count n_arg, uint32_t *tile_sums) {
  __shared__ uint32_t smem[kBlock / kWave];
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + kTile - 1) / kTile;
  for (uint64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t r = 0; r < kTile / kBlock; ++r) {
      const uint64_t i = tile * kTile + r * kBlock + threadIdx.x;
      if (i < n) acc += value(i);
    }
    acc = wave_reduce_sum(acc);
    if (lane_id() == 0) smem[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[tile] = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
  }
}

// One block.  tile_prefix[t] = base + sum_{u<t} tile_sums[u]; totals out.
// mirror_a / mirror_b (optional): 64-bit copies of base + total (e.g. "frontier size" slots).
__global__ __launch_bounds__(kBlock) void k_tile_prefix(const uint32_t *tile_sums, Count n_arg,
                                                        uint32_t *tile_prefix, const uint32_t *base_in,
                                                        uint32_t *total32_out, uint64_t *total64_out,
                                                        uint64_t *mirror_a, uint64_t *mirror_b);

template <typename ValueF, typename EmitF>
__global__ __launch_bounds__(kBlock) void k_tile_apply(ValueF value, EmitF emit, Count n_arg,
                                                       const uint32_t *tile_prefix) {
  __shared__ uint32_t smem[kBlock / kWave];
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + kTile - 1) / kTile;
  for (uint64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    uint32_t running = tile_prefix[tile];
#pragma unroll
    for (uint32_t r = 0; r < kTile / kBlock; ++r) {
      const uint64_t i = tile * kTile + r * kBlock + threadIdx.x;
      const uint32_t v = (i < n) ? value(i) : 0u;
      uint32_t total;
      const uint32_t excl = block_exclusive_scan(v, smem, total);
      if (i < n) emit(i, v, running + excl);
      running += total;
    }
  }
}

// Host helper: run the three phases on `stream`.  scratch: tile_scan_words(n_max) uint32.
template <typename ValueF, typename EmitF>
inline int tile_scan(ValueF value, EmitF emit, size_t n_max, Count n, uint32_t *scratch,
                     const uint32_t *base_in, uint32_t *total32_out, uint64_t *total64_out,
                     hipStream_t stream, uint64_t *mirror_a = nullptr, uint64_t *mirror_b = nullptr) {
  const size_t nt = num_tiles_for(n_max);
  uint32_t *tile_sums = scratch;
  uint32_t *tile_prefix = scratch + nt + 2;
  const int grid = grid_for(nt, 1);
  hipLaunchKernelGGL((k_tile_reduce<ValueF>), dim3(grid), dim3(kBlock), 0, stream, value, n, tile_sums);
  GGMS_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(kBlock), 0, stream, tile_sums, n, tile_prefix, base_in,
                     total32_out, total64_out, mirror_a, mirror_b);
  GGMS_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_tile_apply<ValueF, EmitF>), dim3(grid), dim3(kBlock), 0, stream, value, emit, n,
                     tile_prefix);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // namespace ggms
