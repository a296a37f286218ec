// radix_sort.h -- stable LSD radix sort of (u32 key, u32 value) pairs, 8 bits per pass.
//
// Replaces cub::DeviceRadixSort::SortPairs (cuda_sampling_weighted_khop.cu:172-181).  A stable sort
// is fully determined by its definition, so any correct implementation is result-identical.
// Per pass: (1) per-tile digit histogram, laid out [digit][tile] so that one exclusive scan gives each
// (digit, tile) its global base; (2) the scan (tile_scan over 256 x tiles counters); (3) stable
// scatter: inside a tile, waves own consecutive 64-item chunks; a lane's rank among equal digits of its
// chunk comes from 8 ballots (multi-split), chunk bases from LDS counters in chunk order.
// The element count may live on the device (ggms::Count).
#pragma once

#include "tile_scan.h"

namespace ggms {

constexpr uint32_t kSortTile = 2048; // items per block per pass (8 chunks of 64 per wave, 4 waves)

inline size_t sort_tiles(size_t n) { return (n + kSortTile - 1) / kSortTile; }
// scratch words: histogram [256][tiles] + its scan scratch
inline size_t sort_scratch_words(size_t n) { return 256 * sort_tiles(n) + tile_scan_words(256 * sort_tiles(n)) + 64; }

__global__ __launch_bounds__(kBlock) void k_sort_hist(const uint32_t *__restrict__ keys, Count n_arg, uint32_t shift,
                                                      uint32_t *__restrict__ hist, uint32_t tiles_max) {
  __shared__ uint32_t h[256];
  const uint64_t n = n_arg.get();
  for (uint32_t tile = blockIdx.x; tile < tiles_max; tile += gridDim.x) {
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)tile * kSortTile;
    for (uint32_t i = threadIdx.x; i < kSortTile; i += kBlock)
      if (base + i < n) atomicAdd(&h[(keys[base + i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(uint64_t)threadIdx.x * tiles_max + tile] = h[threadIdx.x]; // zero for tiles beyond n
    __syncthreads();
  }
}

struct HistValue {
  const uint32_t *hist;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return hist[i]; }
};
struct HistStore {
  uint32_t *hist;
  __device__ __forceinline__ void operator()(uint64_t i, uint32_t, uint32_t excl) const { hist[i] = excl; }
};

__global__ __launch_bounds__(kBlock) void k_sort_scatter(const uint32_t *__restrict__ keys_in,
                                                         const uint32_t *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out,
                                                         uint32_t *__restrict__ vals_out, Count n_arg, uint32_t shift,
                                                         const uint32_t *__restrict__ hist, uint32_t tiles_max) {
  constexpr uint32_t WAVES = kBlock / kWave, CHUNKS = kSortTile / kWave / WAVES; // 4 waves x 8 chunks
  __shared__ uint32_t wave_cnt[WAVES][256]; // digits seen so far inside the tile, per wave
  __shared__ uint32_t base[256];
  const uint64_t n = n_arg.get();
  const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
  for (uint32_t tile = blockIdx.x; tile < tiles_max; tile += gridDim.x) {
    const uint64_t tile_base = (uint64_t)tile * kSortTile;
    if (tile_base >= n) break;
    base[threadIdx.x] = hist[(uint64_t)threadIdx.x * tiles_max + tile];
    for (uint32_t w = 0; w < WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    // pass A: per-wave digit counts (wave w owns items [w*CHUNKS*64, (w+1)*CHUNKS*64) of the tile)
    for (uint32_t c = 0; c < CHUNKS; ++c) {
      const uint64_t i = tile_base + (uint64_t)(wave * CHUNKS + c) * kWave + lane;
      if (i < n) atomicAdd(&wave_cnt[wave][(keys_in[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    // exclusive prefix over waves, per digit (thread = digit)
    {
      uint32_t run = 0;
      for (uint32_t w = 0; w < WAVES; ++w) {
        const uint32_t v = wave_cnt[w][threadIdx.x];
        wave_cnt[w][threadIdx.x] = run;
        run += v;
      }
    }
    __syncthreads();
    // pass B: chunks in order; rank inside a chunk by multi-split ballots
    for (uint32_t c = 0; c < CHUNKS; ++c) {
      const uint64_t i = tile_base + (uint64_t)(wave * CHUNKS + c) * kWave + lane;
      const bool valid = i < n;
      const uint32_t key = valid ? keys_in[i] : 0u;
      const uint32_t val = valid ? vals_in[i] : 0u;
      const uint32_t digit = (key >> shift) & 255u;
      uint64_t same = __ballot(valid);
#pragma unroll
      for (uint32_t b = 0; b < 8; ++b) {
        const uint64_t m = __ballot((digit >> b) & 1u);
        same &= ((digit >> b) & 1u) ? m : ~m;
      }
      const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
      uint32_t before = 0;
      if (valid) before = wave_cnt[wave][digit];
      __builtin_amdgcn_wave_barrier();
      if (valid) {
        const uint64_t dst = (uint64_t)base[digit] + before + rank;
        keys_out[dst] = key;
        vals_out[dst] = val;
        if (rank + 1 == (uint32_t)__popcll(same)) wave_cnt[wave][digit] = before + rank + 1; // last of its digit
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
}

// keys/vals ping-pong between (k0,v0) and (k1,v1).  `passes` bytes of the key are sorted (4 = the whole key; fewer
// when the caller knows the keys' range -- the rest of the key must then agree wherever the sorted bytes do);
// *in_second tells whether the result ended in (k1,v1) (odd number of passes).
inline int radix_sort_pairs(uint32_t *k0, uint32_t *v0, uint32_t *k1, uint32_t *v1, size_t n_max, Count n,
                            uint32_t *scratch, hipStream_t s, uint32_t passes = 4, bool *in_second = nullptr) {
  if (in_second) *in_second = (passes & 1u) != 0;
  if (n_max == 0) return GGMS_OK;
  const uint32_t tiles = (uint32_t)sort_tiles(n_max);
  uint32_t *hist = scratch;
  uint32_t *scan_scratch = scratch + 256 * (size_t)tiles;
  const int grid = grid_for(tiles, 1);
  for (uint32_t pass = 0; pass < passes; ++pass) {
    const uint32_t shift = 8 * pass;
    uint32_t *ki = (pass & 1) ? k1 : k0, *vi = (pass & 1) ? v1 : v0;
    uint32_t *ko = (pass & 1) ? k0 : k1, *vo = (pass & 1) ? v0 : v1;
    hipLaunchKernelGGL(k_sort_hist, dim3(grid), dim3(kBlock), 0, s, ki, n, shift, hist, tiles);
    GGMS_LAUNCH_CHECK();
    int rc = tile_scan(HistValue{hist}, HistStore{hist}, 256 * (size_t)tiles, count_of(256 * (size_t)tiles),
                       ScanArea{scan_scratch, pass != 0}, nullptr, nullptr, nullptr, s); // one clear for all passes
    if (rc != GGMS_OK) return rc;
    hipLaunchKernelGGL(k_sort_scatter, dim3(grid), dim3(kBlock), 0, s, ki, vi, ko, vo, n, shift, hist, tiles);
    GGMS_LAUNCH_CHECK();
  }
  return GGMS_OK;
}

} // namespace ggms
