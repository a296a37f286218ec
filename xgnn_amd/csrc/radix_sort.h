// radix_sort.h -- stable LSD radix sort of (u32 key, u32 value) pairs.
//
// Replaces cub::DeviceRadixSort::SortPairs (cuda_sampling_weighted_khop.cu:172-181).  A stable sort
// is fully determined by its definition, so any correct implementation is result-identical.
//
// What is sorted here is a frontier's seed positions by node id: 8 000 .. a few 100 000 pairs, so every launch
// sits near the launch-latency floor and the number of launches is what counts.
//   * up to kSmallSort pairs: ONE workgroup sorts them in LDS, all passes in one launch (k_sort_small);
//   * beyond: per pass (1) per-tile digit histogram, laid out [digit][tile] so that one exclusive scan gives each
//     (digit, tile) its global base; (2) the scan (tile_scan over radix x tiles counters); (3) stable scatter:
//     inside a tile, waves own consecutive 64-item chunks; a lane's rank among equal digits of its chunk comes
//     from one ballot per digit bit (multi-split), chunk bases from LDS counters in chunk order.  Digits are 11
//     bits wide (two passes cover ids below 2^22, three any 32-bit key) up to kWideDigitItems pairs -- fewer,
//     wider passes; larger inputs keep 8-bit digits, whose scatter writes in longer runs.
// The caller states the key range (`key_limit`: real keys are below it; the empty key, all ones, sorts last because
// the passes cover every bit a real key can have set -- see radix_plan).
// The element count may live on the device (ggms::Count).
#pragma once

#include "tile_scan.h"

namespace ggms {

constexpr uint32_t kSortTile = 2048;          // items per block per pass (8 chunks of 64 per wave, 4 waves)
constexpr uint32_t kSmallSort = 8192;         // pairs one workgroup sorts in LDS
constexpr size_t kWideDigitItems = 1u << 20;  // 11-bit digits up to this many pairs

inline size_t sort_tiles(size_t n) { return (n + kSortTile - 1) / kSortTile; }

struct RadixPlan {
  uint32_t bits, passes;
};
// key_limit = 0: any 32-bit key.  Keys below key_limit sort correctly once the passes cover all their bits; the empty
// key (all ones) then still compares above them as long as key_limit <= 2^(bits * passes) - 1, i.e. no real key has
// all covered bits set.
inline RadixPlan radix_plan(size_t n_max, uint32_t key_limit) {
  const uint32_t bits = n_max <= kWideDigitItems ? 11u : 8u;
  uint32_t passes = (32 + bits - 1) / bits;
  if (key_limit)
    for (uint32_t p = 1; p < passes; ++p)
      if (bits * p < 32 && (unsigned long long)key_limit <= (1ull << (bits * p)) - 1ull) { passes = p; break; }
  return {bits, passes};
}
// scratch words: histogram [radix][tiles] + its scan scratch (sized for the wider digit)
inline size_t sort_scratch_words(size_t n) {
  if (n <= kSmallSort) return 64;
  const size_t radix = n <= kWideDigitItems ? 2048 : 256;
  return radix * sort_tiles(n) + tile_scan_words(radix * sort_tiles(n)) + 64;
}

template <uint32_t BITS>
__global__ __launch_bounds__(kBlock) void k_sort_hist(const uint32_t *__restrict__ keys, Count n_arg, uint32_t shift,
                                                      uint32_t *__restrict__ hist, uint32_t tiles_max) {
  constexpr uint32_t R = 1u << BITS;
  __shared__ uint32_t h[R];
  const uint64_t n = n_arg.get();
  for (uint32_t tile = blockIdx.x; tile < tiles_max; tile += gridDim.x) {
    for (uint32_t d = threadIdx.x; d < R; d += kBlock) h[d] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)tile * kSortTile;
    for (uint32_t i = threadIdx.x; i < kSortTile; i += kBlock)
      if (base + i < n) atomicAdd(&h[(keys[base + i] >> shift) & (R - 1)], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < R; d += kBlock) hist[(uint64_t)d * tiles_max + tile] = h[d]; // zero beyond n
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

// lanes of the wave whose (valid) digit equals mine: one ballot per digit bit
template <uint32_t BITS>
__device__ __forceinline__ uint64_t same_digit_mask(uint32_t digit, bool valid) {
  uint64_t same = __ballot(valid);
#pragma unroll
  for (uint32_t b = 0; b < BITS; ++b) {
    const uint64_t m = __ballot((digit >> b) & 1u);
    same &= ((digit >> b) & 1u) ? m : ~m;
  }
  return same;
}

template <uint32_t BITS>
__global__ __launch_bounds__(kBlock) void k_sort_scatter(const uint32_t *__restrict__ keys_in,
                                                         const uint32_t *__restrict__ vals_in,
                                                         uint32_t *__restrict__ keys_out,
                                                         uint32_t *__restrict__ vals_out, Count n_arg, uint32_t shift,
                                                         const uint32_t *__restrict__ hist, uint32_t tiles_max) {
  constexpr uint32_t R = 1u << BITS;
  constexpr uint32_t WAVES = kBlock / kWave, CHUNKS = kSortTile / kWave / WAVES; // 4 waves x 8 chunks
  __shared__ uint32_t wave_cnt[WAVES][R]; // digits seen so far inside the tile, per wave
  __shared__ uint32_t base[R];
  const uint64_t n = n_arg.get();
  const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
  for (uint32_t tile = blockIdx.x; tile < tiles_max; tile += gridDim.x) {
    const uint64_t tile_base = (uint64_t)tile * kSortTile;
    if (tile_base >= n) break;
    for (uint32_t d = threadIdx.x; d < R; d += kBlock) {
      base[d] = hist[(uint64_t)d * tiles_max + tile];
      for (uint32_t w = 0; w < WAVES; ++w) wave_cnt[w][d] = 0;
    }
    __syncthreads();
    // pass A: per-wave digit counts (wave w owns items [w*CHUNKS*64, (w+1)*CHUNKS*64) of the tile)
    for (uint32_t c = 0; c < CHUNKS; ++c) {
      const uint64_t i = tile_base + (uint64_t)(wave * CHUNKS + c) * kWave + lane;
      if (i < n) atomicAdd(&wave_cnt[wave][(keys_in[i] >> shift) & (R - 1)], 1u);
    }
    __syncthreads();
    // exclusive prefix over waves, per digit
    for (uint32_t d = threadIdx.x; d < R; d += kBlock) {
      uint32_t run = 0;
      for (uint32_t w = 0; w < WAVES; ++w) {
        const uint32_t v = wave_cnt[w][d];
        wave_cnt[w][d] = run;
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
      const uint32_t digit = (key >> shift) & (R - 1);
      const uint64_t same = same_digit_mask<BITS>(digit, valid);
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

// Up to kSmallSort pairs: one 1024-thread workgroup, the pairs ping-pong between two LDS buffers, 8-bit digits.
// The same stable scatter as above with the workgroup as the only tile: wave w owns items [512 w, 512 w + 512).
__global__ __launch_bounds__(1024) void k_sort_small(const uint32_t *__restrict__ keys_in,
                                                     const uint32_t *__restrict__ vals_in,
                                                     uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                     Count n_arg, uint32_t passes) {
  constexpr uint32_t WAVES = 16, CHUNKS = kSmallSort / kWave / WAVES; // 16 waves x 8 chunks
  extern __shared__ uint32_t sort_lds[]; // keys[2][kSmallSort], vals[2][kSmallSort]
  __shared__ uint32_t wave_cnt[WAVES][256];
  __shared__ uint32_t base[256];
  uint32_t *const kbuf = sort_lds, *const vbuf = sort_lds + 2 * kSmallSort;
  const uint32_t n = (uint32_t)n_arg.get();
  const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
  for (uint32_t i = threadIdx.x; i < n; i += 1024) {
    kbuf[i] = keys_in[i];
    vbuf[i] = vals_in[i];
  }
  uint32_t cur = 0;
  for (uint32_t pass = 0; pass < passes; ++pass, cur ^= 1u) {
    const uint32_t shift = 8 * pass;
    const uint32_t *ki = kbuf + cur * kSmallSort, *vi = vbuf + cur * kSmallSort;
    uint32_t *ko = kbuf + (cur ^ 1u) * kSmallSort, *vo = vbuf + (cur ^ 1u) * kSmallSort;
    if (threadIdx.x < 256)
      for (uint32_t w = 0; w < WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;
    __syncthreads(); // also: the loads above / the previous pass's scatter are complete
    for (uint32_t c = 0; c < CHUNKS; ++c) {
      const uint32_t i = (wave * CHUNKS + c) * kWave + lane;
      if (i < n) atomicAdd(&wave_cnt[wave][(ki[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) { // exclusive prefix over waves per digit, then over digits (4 waves of 64 digits)
      uint32_t run = 0;
      for (uint32_t w = 0; w < WAVES; ++w) {
        const uint32_t v = wave_cnt[w][threadIdx.x];
        wave_cnt[w][threadIdx.x] = run;
        run += v;
      }
      base[threadIdx.x] = run; // digit total
    }
    __syncthreads();
    if (threadIdx.x < kWave) { // 64 lanes x 4 consecutive digits
      uint32_t t[4], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) { t[k] = base[4 * lane + k]; sum += t[k]; }
      uint32_t run = wave_inclusive_scan(sum) - sum;
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) { base[4 * lane + k] = run; run += t[k]; }
    }
    __syncthreads();
    for (uint32_t c = 0; c < CHUNKS; ++c) {
      const uint32_t i = (wave * CHUNKS + c) * kWave + lane;
      const bool valid = i < n;
      const uint32_t key = valid ? ki[i] : 0u;
      const uint32_t val = valid ? vi[i] : 0u;
      const uint32_t digit = (key >> shift) & 255u;
      const uint64_t same = same_digit_mask<8>(digit, valid);
      const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
      uint32_t before = 0;
      if (valid) before = wave_cnt[wave][digit];
      __builtin_amdgcn_wave_barrier();
      if (valid) {
        const uint32_t dst = base[digit] + before + rank;
        ko[dst] = key;
        vo[dst] = val;
        if (rank + 1 == (uint32_t)__popcll(same)) wave_cnt[wave][digit] = before + rank + 1;
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < n; i += 1024) {
    keys_out[i] = kbuf[cur * kSmallSort + i];
    vals_out[i] = vbuf[cur * kSmallSort + i];
  }
}

template <uint32_t BITS>
inline int radix_sort_passes(uint32_t *k0, uint32_t *v0, uint32_t *k1, uint32_t *v1, size_t n_max, Count n,
                             uint32_t *scratch, hipStream_t s, uint32_t passes) {
  constexpr size_t R = 1u << BITS;
  const uint32_t tiles = (uint32_t)sort_tiles(n_max);
  uint32_t *hist = scratch;
  uint32_t *scan_scratch = scratch + R * (size_t)tiles;
  const int grid = grid_for(tiles, 1);
  for (uint32_t pass = 0; pass < passes; ++pass) {
    const uint32_t shift = BITS * pass;
    uint32_t *ki = (pass & 1) ? k1 : k0, *vi = (pass & 1) ? v1 : v0;
    uint32_t *ko = (pass & 1) ? k0 : k1, *vo = (pass & 1) ? v0 : v1;
    hipLaunchKernelGGL(k_sort_hist<BITS>, dim3(grid), dim3(kBlock), 0, s, ki, n, shift, hist, tiles);
    GGMS_LAUNCH_CHECK();
    int rc = tile_scan(HistValue{hist}, HistStore{hist}, R * (size_t)tiles, count_of(R * (size_t)tiles),
                       ScanArea{scan_scratch, pass != 0}, nullptr, nullptr, nullptr, s); // one clear for all passes
    if (rc != GGMS_OK) return rc;
    hipLaunchKernelGGL(k_sort_scatter<BITS>, dim3(grid), dim3(kBlock), 0, s, ki, vi, ko, vo, n, shift, hist, tiles);
    GGMS_LAUNCH_CHECK();
  }
  return GGMS_OK;
}

// keys/vals start in (k0,v0); *in_second tells whether the result ended in (k1,v1).  key_limit: see radix_plan.
inline int radix_sort_pairs(uint32_t *k0, uint32_t *v0, uint32_t *k1, uint32_t *v1, size_t n_max, Count n,
                            uint32_t *scratch, hipStream_t s, uint32_t key_limit, bool *in_second) {
  if (n_max == 0) {
    *in_second = false;
    return GGMS_OK;
  }
  if (n_max <= kSmallSort) {
    uint32_t passes = 4; // 8-bit digits here
    for (uint32_t p = 1; p < 4; ++p)
      if (key_limit && (unsigned long long)key_limit <= (1ull << (8 * p)) - 1ull) { passes = p; break; }
    static const int raised = raise_dynamic_lds(reinterpret_cast<const void *>(&k_sort_small),
                                                4 * kSmallSort * sizeof(uint32_t), "k_sort_small");
    if (raised != GGMS_OK) return raised;
    hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(1024), 4 * kSmallSort * sizeof(uint32_t), s, k0, v0, k1, v1, n, passes);
    GGMS_LAUNCH_CHECK();
    *in_second = true;
    return GGMS_OK;
  }
  const RadixPlan plan = radix_plan(n_max, key_limit);
  *in_second = (plan.passes & 1u) != 0;
  return plan.bits == 11 ? radix_sort_passes<11>(k0, v0, k1, v1, n_max, n, scratch, s, plan.passes)
                         : radix_sort_passes<8>(k0, v0, k1, v1, n_max, n, scratch, s, plan.passes);
}

} // namespace ggms
