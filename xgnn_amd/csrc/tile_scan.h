// tile_scan.h -- ordered (stable) device-wide exclusive scan / compaction, count on the device.
//
// Replaces the reference's {count kernel, cub::DeviceScan::ExclusiveSum over grid+1 entries, compact
// kernel with cub::BlockScan} triple (e.g. cuda_sampling_khop3.cu:148-230,286-295; cuda_hashtable.cu:
// 197-232,406-458; cuda_cache_manager_device.cu:40-169) and its intermediate host syncs.
// Two interchangeable forms (same results, same scratch): one launch with decoupled look-back for inputs of up
// to kSinglePassTiles tiles (every block is resident, the look-back is a handful of descriptor reads and two
// launch latencies are saved), three launches beyond that (see scan_three_pass).
//
// Single pass, decoupled look-back:
//   * tiles of 1024 items (4 rounds of 256 threads, item = tile*1024 + round*256 + thread -- the
//     reference's kCudaTileSize/kCudaBlockSize mapping, so output order = input order);
//   * tile = workgroup id; nothing is assumed about when a predecessor's workgroup runs -- a look-back that has
//     waited long enough computes the missing aggregate itself (scan_lookback's Help);
//   * per tile one 64-bit descriptor {epoch : 30 | flag : 2 | value : 32}: flag A = tile aggregate,
//     P = inclusive prefix.  Data and tag travel in ONE 8-byte relaxed agent-scope atomic store / load
//     (no separate flag, hence no release/acquire pair to get wrong across XCDs); a word whose epoch is
//     not the launch's epoch is "not yet written", so descriptors are never cleared between launches;
//   * the element count may live in device memory (ggms::Count): no host round trip.
// Callers clear the control + descriptor region once (hipMemsetAsync) per API call / per batch.
#pragma once

#include <atomic>
#include <cstdlib>

#include "ggms_device.h"

namespace ggms {

constexpr uint32_t kTile = 1024;

inline size_t num_tiles_for(size_t n) { return (n + kTile - 1) / kTile; }
// scratch in uint32 words: 8 control words, one 64-bit descriptor per tile (single pass), tile sums + prefixes
// (three launches); the two forms keep their words apart so that scans of both kinds can share one area
inline size_t tile_scan_words(size_t n) { return 12 + 4 * (num_tiles_for(n) + 1); }
// inputs of at most this many tiles take the single-pass kernel
constexpr size_t kSinglePassTiles = 256;
// words (from the aligned start of an area) that must be zero before the first single-pass scan of a batch
inline size_t scan_clear_words(size_t n_max) {
  const size_t nt = num_tiles_for(n_max);
  return 8 + 2 * ((nt < kSinglePassTiles ? nt : kSinglePassTiles) + 1);
}

// where a scan keeps its control words + descriptors; `cleared` = the caller zeroed it already (one
// memset per batch instead of one per scan; descriptors are epoch-tagged, so scans may share an area)
struct ScanArea {
  uint32_t *words;
  bool cleared;
  uint32_t *stash = nullptr; // optional n_max words: value(i) is evaluated ONCE (reduce pass) and re-read from here
  uint32_t *chunk = nullptr; // optional chunk_desc_words() words for the chunked owner scan (hashtable.hip); zeroed
                             // with the area (`cleared`), never shared with the 64-bit descriptors of the other scans
  uint32_t *tickets = nullptr; // optional kTicketSets ticket sets (below), zeroed with the area; a ticketed sampler
  uint32_t next_ticket_set = 0; // launch of the batch takes the next one (take_ticket_set)
  uint32_t *status = nullptr;   // where the scans of this area report a bound they hit: the batch's own status word;
                                // NULL = the device's sticky word (leaf operators)
  uint32_t *status_word() const { return status ? status : device_status_word(); }
};

// ---- tickets without a hot word ----------------------------------------------------------------------------------
// Atomics on ONE address are served one after the other at the memory side: 11 ns each, returning or not
// (tools/micro_ticket.hip, profiles/r03_micro_ticket.txt).  A layer of 800 K seeds is 6250 tiles: tickets from a single
// counter, one failing ticket per workgroup at the end and a "last one out re-arms the counter" atomic on the
// neighbouring word come to 10 K atomics on one line = 114 us of a 200-us kernel.  So the tickets of a launch come from
// kTicketLanes counters 64 bytes apart: workgroup w draws from lane w % S, lane l hands out the tiles l, l + S,
// l + 2 S, ...  A tile's predecessors are then no longer guaranteed to be running when it looks back (another lane's
// next ticket may belong to a workgroup that has not started, and a second kernel's waiting workgroups may hold the
// slot it needs): the look-backs of these kernels therefore compute a predecessor that does not show up for
// themselves (scan_lookback's Help).  Nothing is re-armed: every ticketed launch of a batch has a ticket set of its
// own, all of them zeroed by the batch prologue.
constexpr uint32_t kTicketLanes = 32, kTicketStride = 16;
constexpr uint32_t kTicketWords = kTicketLanes * kTicketStride; // one set
constexpr uint32_t kTicketSets = 20;                             // per batch: one per ticketed sampler launch (<= 16 layers)
// strict: ONE counter -- a taken tile's predecessors are then always running or done (the forward-progress rule that
// needs nothing else), at 11 ns per ticket: for kernels whose look-back cannot compute a predecessor for itself.
__device__ __forceinline__ uint64_t take_ticket(uint32_t *set, bool strict = false) { // one thread of the workgroup
  const uint32_t lanes = strict ? 1u : (gridDim.x < kTicketLanes ? gridDim.x : kTicketLanes);
  const uint32_t l = blockIdx.x % lanes;
  return (uint64_t)l + (uint64_t)lanes * atomicAdd(&set[kTicketStride * l], 1u);
}
inline uint32_t *take_ticket_set(ScanArea *a) {
  if (!a || !a->tickets || a->next_ticket_set >= kTicketSets) return nullptr;
  return a->tickets + (size_t)kTicketWords * a->next_ticket_set++;
}

inline uint32_t next_scan_epoch() {
  static std::atomic<uint32_t> g{0};
  uint32_t e;
  do { e = (g.fetch_add(1) + 1) & 0x3fffffffu; } while (e == 0);
  return e;
}

__device__ __forceinline__ unsigned long long scan_desc(uint32_t epoch, uint32_t flag, uint32_t value) {
  return ((unsigned long long)((epoch << 2) | flag) << 32) | value;
}

// ---- a wait that does not depend on where or when another workgroup runs --------------------------------------------
// HIP promises nothing about dispatch order or placement, and a workgroup that spins holds its slot: two kernels with
// look-back chains on one device (two processes sharing a GPU, two batches in flight) can each fill the slots the
// other's next workgroup needs, and then every resident workgroup waits for one that cannot start.  (Seen: two bench
// ranks on one MI355X at papers100M size, GGMS_STATUS_SCAN_SPIN in half of the runs.)  So a look-back only waits
// `patience` polls for a predecessor's descriptor; after that it COMPUTES the missing aggregates itself from the
// scan's input (Help: tile -> what the tile's descriptor would hold; 64 lanes, result uniform) and publishes them.
// Every resident workgroup therefore finishes whatever the others do.  The computed word is only used if the tile's
// own word is still missing AFTER the inputs were read (its owner publishes before it overwrites anything: scans
// that work in place stay exact).  NoHelp keeps the plain bounded wait (strictly ticketed kernels: a taken tile's
// predecessors are always running).
struct NoHelp {
  static constexpr bool kCan = false;
  __device__ __forceinline__ uint32_t operator()(uint64_t) const { return 0u; }
};
constexpr uint32_t kNoPatienceLimit = 0xffffffffu;
inline std::atomic<uint32_t> &scan_patience_word() {
  static std::atomic<uint32_t> v{2048u};
  return v;
}
inline uint32_t scan_patience() { return scan_patience_word().load(std::memory_order_relaxed); }
// ggms_debug_delay_next_scan (tests): the workgroup of tile / chunk 0 of the next single-pass scan "starts late"
inline std::atomic<uint32_t> &scan_delay_word() {
  static std::atomic<uint32_t> v{0u};
  return v;
}

// Decoupled look-back, run by the 64 lanes of ONE wave: sum of the aggregates of the tiles before `tile`, back to
// the nearest published inclusive prefix (which carries `base` in from tile 0).  64 predecessors per step.
// Bounded: a protocol error must not hang the GPU -- after 2^22 polls the wave gives up, ORs kErrScanSpin into
// the device status word (the host fails the batch on it) and returns what it has.
template <typename Help = NoHelp>
__device__ __forceinline__ uint32_t scan_lookback(unsigned long long *desc, uint64_t tile, uint32_t epoch,
                                                  uint32_t *err, uint32_t patience = kNoPatienceLimit,
                                                  const Help &help = Help()) {
  constexpr uint32_t FLAG_A = 1, FLAG_P = 2;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t acc = 0;
  int64_t start = (int64_t)tile - 1; // nearest predecessor is read by lane 0
  for (uint32_t spins = 0;; ++spins) {
    if (spins > (1u << 22)) {
      if (lane == 0 && err) atomicOr(err, kErrScanSpin);
      break;
    }
    const int64_t t = start - (int64_t)lane;
    unsigned long long d = scan_desc(epoch, FLAG_P, 0); // lanes past tile 0 read as "prefix 0"
    if (t >= 0) d = __hip_atomic_load(&desc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t tag = (uint32_t)(d >> 32);
    bool ready = (tag >> 2) == epoch && (tag & 3u) != 0;
    uint64_t pmask = __ballot(ready && (tag & 3u) == FLAG_P);
    const uint64_t rmask = __ballot(ready);
    uint32_t first_p = pmask ? (uint32_t)__builtin_ctzll(pmask) : 64u;
    const uint64_t needed = first_p < 64u ? ((first_p == 63u) ? ~0ull : ((1ull << (first_p + 1)) - 1ull)) : ~0ull;
    if ((rmask & needed) != needed) { // a descriptor between us and the nearest prefix is not written yet
      if (!Help::kCan || spins < patience) {
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      uint64_t missing = needed & ~rmask; // uniform; lanes past tile 0 are ready by construction
      while (missing) {
        const uint32_t bpos = (uint32_t)__builtin_ctzll(missing);
        missing &= missing - 1ull;
        const uint64_t tt = (uint64_t)(start - (int64_t)bpos);
        const uint32_t h = help(tt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the inputs were READ before the word is looked at again
        const unsigned long long now = __hip_atomic_load(&desc[tt], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t ntag = (uint32_t)(now >> 32);
        const bool there = (ntag >> 2) == epoch && (ntag & 3u) != 0;
        const unsigned long long mine = scan_desc(epoch, tt == 0 ? FLAG_P : FLAG_A, h);
        if (!there && lane == 0) __hip_atomic_store(&desc[tt], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the word is OUT before this wave goes on: if it is the last tile, the total it writes (over a count that tile
        // 0's word was computed from) must not become visible before that word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == bpos) d = there ? now : mine;
      }
      tag = (uint32_t)(d >> 32);
      ready = (tag >> 2) == epoch && (tag & 3u) != 0;
      pmask = __ballot(ready && (tag & 3u) == FLAG_P);
      first_p = pmask ? (uint32_t)__builtin_ctzll(pmask) : 64u;
    }
    const uint32_t mine = (lane <= first_p || first_p == 64u) ? (uint32_t)d : 0u;
    acc += wave_reduce_sum(mine);
    if (first_p < 64u) break;
    start -= kWave;
  }
  return acc;
}

// what a sampler kernel that scans its own tiles (k_khop3_fused, k_walk_topk_emit) needs of a scan area
struct FusedScan {
  uint32_t *tick;            // this launch's ticket set (kTicketWords zeroed words), see take_ticket
  unsigned long long *desc;  // one descriptor per tile
  uint32_t epoch;
  uint64_t *num_out;         // total number of edges
  uint32_t *err;
  uint32_t patience;         // polls before a look-back serves itself (scan_patience())
};

// what tile t's descriptor holds, computed by one wave from the scan's input (scan_lookback's Help)
template <typename ValueF>
struct TileSumHelp {
  static constexpr bool kCan = true;
  const ValueF &value;
  uint64_t n;
  uint32_t base;
  __device__ __forceinline__ uint32_t operator()(uint64_t t) const {
    uint32_t acc = 0;
    for (uint32_t k = 0; k < kTile / kWave; ++k) {
      const uint64_t i = t * kTile + k * kWave + (threadIdx.x & 63u);
      if (i < n) acc += value(i);
    }
    return wave_reduce_sum(acc) + (t == 0 ? base : 0u);
  }
};

template <typename ValueF, typename EmitF>
__global__ __launch_bounds__(kBlock) void k_tile_scan(ValueF value, EmitF emit, Count n_arg,
                                                      unsigned long long *desc, uint32_t epoch,
                                                      const uint32_t *base_in, uint32_t *total32_out,
                                                      uint64_t *total64_out, uint64_t *mirror_a, uint64_t *mirror_b,
                                                      uint32_t *err, uint32_t patience, uint32_t delay0) {
  constexpr uint32_t ROUNDS = kTile / kBlock, FLAG_A = 1, FLAG_P = 2;
  __shared__ uint32_t smem[kBlock / kWave];
  __shared__ uint32_t s_prefix;
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + kTile - 1) / kTile;
  // total32_out may alias base_in (a running count updated in place): the last tile overwrites it, but only after
  // tile 0's descriptor is out -- so whoever finds that descriptor missing AFTER this read has read the old value
  // (tile 0's owner and the look-backs that compute tile 0 for themselves check exactly that)
  if (delay0 && blockIdx.x == 0) // test aid: tile 0's owner starts after the others computed its word and the total is out
    for (uint32_t i = 0; i < delay0; ++i) __builtin_amdgcn_s_sleep(127);
  const uint32_t base_seen = base_in ? *base_in : 0u;
  // tile = workgroup id (the launch has one workgroup per tile, at most kSinglePassTiles): a ticket and an exit
  // count from one word are 2 x 256 same-address atomics, 11 ns each (tools/micro_ticket.hip) -- most of an 8-us launch.
  // A predecessor that has not started (nothing is promised about dispatch order) is computed by whoever waits for it
  // (scan_lookback, TileSumHelp).
  for (uint64_t round = 0;; ++round) {
    const uint64_t tile = (uint64_t)blockIdx.x + round * gridDim.x;
    if (tile >= num_tiles) break;
    uint32_t v[ROUNDS], excl[ROUNDS];
    uint32_t running = 0;
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
      const uint64_t i = tile * kTile + r * kBlock + threadIdx.x;
      v[r] = (i < n) ? value(i) : 0u;
      uint32_t total;
      excl[r] = running + block_exclusive_scan(v[r], smem, total);
      running += total;
    }
    if (threadIdx.x < kWave) { // wave 0 publishes and looks back, 64 predecessors per step
      const uint32_t lane = threadIdx.x;
      uint32_t prefix = base_seen;
      if (tile == 0) {
        if (base_in) { // somebody who could not wait may have computed this tile's word already (and the total may be out)
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          const unsigned long long d0 = __hip_atomic_load(&desc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(d0 >> 34) == epoch && ((uint32_t)(d0 >> 32) & 3u) == FLAG_P) prefix = (uint32_t)d0 - running;
        }
        if (lane == 0)
          __hip_atomic_store(&desc[0], scan_desc(epoch, FLAG_P, prefix + running), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // "the owner publishes before it overwrites anything": the word is OUT, device-wide, before the barrier below lets
        // any wave of this workgroup emit -- a scan that works in place (the hashed table's owner flags) rewrites what a
        // helper's value() reads, and a helper that still finds the word missing must have read the old input.  A
        // workgroup-scope barrier alone does not wait for the store (XCD L2s are not coherent); one fence per launch.
        __threadfence();
      } else {
        if (lane == 0)
          __hip_atomic_store(&desc[tile], scan_desc(epoch, FLAG_A, running), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t acc = scan_lookback(desc, tile, epoch, err, patience, TileSumHelp<ValueF>{value, n, base_seen});
        prefix = acc; // already includes `base` through tile 0's inclusive prefix
        if (lane == 0)
          __hip_atomic_store(&desc[tile], scan_desc(epoch, FLAG_P, prefix + running), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        s_prefix = prefix;
        if (tile + 1 == num_tiles) { // the last tile knows the grand total
          const uint32_t total = prefix + running;
          if (total32_out) *total32_out = total;
          if (total64_out) *total64_out = (uint64_t)(total - base_seen); // nobody but this tile overwrites base_in
          if (mirror_a) *mirror_a = (uint64_t)total;
          if (mirror_b) *mirror_b = (uint64_t)total;
        }
      }
    }
    __syncthreads();
    const uint32_t prefix = s_prefix;
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
      const uint64_t i = tile * kTile + r * kBlock + threadIdx.x;
      if (i < n) emit(i, v[r], prefix + excl[r]);
    }
    __syncthreads(); // s_prefix is rewritten next iteration
  }
  if (threadIdx.x == 0 && num_tiles == 0 && blockIdx.x == 0) { // empty input: totals = base
    if (total32_out) *total32_out = base_seen;
    if (total64_out) *total64_out = 0;
    if (mirror_a) *mirror_a = (uint64_t)base_seen;
    if (mirror_b) *mirror_b = (uint64_t)base_seen;
  }
}

// ---- three-launch variant (reduce / prefix / apply): the default ----
template <typename ValueF>
__global__ __launch_bounds__(kBlock) void k_tile_reduce(ValueF value, Count n_arg, uint32_t *tile_sums,
                                                        uint32_t *__restrict__ stash) {
  __shared__ uint32_t smem[kBlock / kWave];
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + kTile - 1) / kTile;
  for (uint64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t r = 0; r < kTile / kBlock; ++r) {
      const uint64_t i = tile * kTile + r * kBlock + threadIdx.x;
      if (i < n) {
        const uint32_t v = value(i);
        if (stash) stash[i] = v;
        acc += v;
      }
    }
    acc = wave_reduce_sum(acc);
    if (lane_id() == 0) smem[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[tile] = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
  }
}


template <typename ValueF, typename EmitF>
__global__ __launch_bounds__(kBlock) void k_tile_apply(ValueF value, EmitF emit, Count n_arg,
                                                       const uint32_t *tile_prefix,
                                                       const uint32_t *__restrict__ stash) {
  __shared__ uint32_t smem[kBlock / kWave];
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + kTile - 1) / kTile;
  for (uint64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    uint32_t running = tile_prefix[tile];
#pragma unroll
    for (uint32_t r = 0; r < kTile / kBlock; ++r) {
      const uint64_t i = tile * kTile + r * kBlock + threadIdx.x;
      const uint32_t v = (i < n) ? (stash ? stash[i] : value(i)) : 0u;
      uint32_t total;
      const uint32_t excl = block_exclusive_scan(v, smem, total);
      if (i < n) emit(i, v, running + excl);
      running += total;
    }
  }
}

// Tiny inputs (n_max <= kSmallScan): ONE block, one launch instead of three.  Measured on MI355X: a single
// 1024-thread block walking 10 K items costs ~20 us against ~15 us for the three tiny launches, so the
// threshold stays at one chunk.
constexpr uint32_t kSmallScan = 1024;
template <typename ValueF, typename EmitF>
__global__ __launch_bounds__(1024) void k_small_scan(ValueF value, EmitF emit, Count n_arg, const uint32_t *base_in,
                                                     uint32_t *total32_out, uint64_t *total64_out, uint64_t *mirror_a,
                                                     uint64_t *mirror_b) {
  constexpr uint32_t CHUNKS = kSmallScan / 1024;
  __shared__ uint32_t wsum[2][16];
  const uint64_t n = n_arg.get();
  const uint32_t base = base_in ? *base_in : 0u;
  const uint32_t w = threadIdx.x >> 6;
  uint32_t v[CHUNKS];
#pragma unroll
  for (uint32_t c = 0; c < CHUNKS; ++c) {
    const uint64_t i = (uint64_t)c * 1024 + threadIdx.x;
    v[c] = (i < n) ? value(i) : 0u;
  }
  uint32_t running = base;
#pragma unroll
  for (uint32_t c = 0; c < CHUNKS; ++c) {
    if ((uint64_t)c * 1024 >= n) break; // uniform
    const uint64_t i = (uint64_t)c * 1024 + threadIdx.x;
    const uint32_t incl = wave_inclusive_scan(v[c]);
    if (lane_id() == 63) wsum[c & 1][w] = incl;
    __syncthreads(); // double-buffered wsum: one barrier per chunk
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
      const uint32_t sw = wsum[c & 1][k];
      if (k < w) before += sw;
      total += sw;
    }
    if (i < n) emit(i, v[c], running + before + incl - v[c]);
    running += total;
  }
  if (threadIdx.x == 0) {
    if (total32_out) *total32_out = running;
    if (total64_out) *total64_out = (uint64_t)(running - base);
    if (mirror_a) *mirror_a = (uint64_t)running;
    if (mirror_b) *mirror_b = (uint64_t)running;
  }
}

// Host helper: run the three phases on `stream`.  scratch: tile_scan_words(n_max) uint32.

template <int DUMMY>
__global__ __launch_bounds__(kBlock) void k_tile_prefix_t(const uint32_t *tile_sums, Count n_arg,
                                                        uint32_t *tile_prefix, const uint32_t *base_in,
                                                        uint32_t *total32_out, uint64_t *total64_out,
                                                        uint64_t *mirror_a, uint64_t *mirror_b) {
  __shared__ uint32_t smem[kBlock / kWave];
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + kTile - 1) / kTile;
  const uint32_t base = base_in ? *base_in : 0u;
  uint32_t running = base;
  for (uint64_t t0 = 0; t0 < num_tiles; t0 += kBlock) {
    const uint64_t t = t0 + threadIdx.x;
    const uint32_t v = (t < num_tiles) ? tile_sums[t] : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(v, smem, total);
    if (t < num_tiles) tile_prefix[t] = running + excl;
    running += total;
  }
  if (threadIdx.x == 0) {
    tile_prefix[num_tiles] = running;
    if (total32_out) *total32_out = running;
    if (total64_out) *total64_out = (uint64_t)(running - base);
    if (mirror_a) *mirror_a = (uint64_t)running;
    if (mirror_b) *mirror_b = (uint64_t)running;
  }
}


// Large inputs: three launches.  A/B on MI355X (bench.py, same box): 0.403 ms vs 0.414 ms of sampling per batch
// with the single-pass kernel everywhere -- at ~900 tiles its 1-tile-per-block latency chain costs what the two
// extra launches cost -- and the three-launch form has no inter-workgroup wait at all.  Small inputs are the
// opposite case: every launch sits on the ~5 us floor, so one launch beats three.
inline bool scan_single_pass(size_t n_max) { return num_tiles_for(n_max) <= kSinglePassTiles; }

// Host helper.  scratch: tile_scan_words(n_max) uint32, 8-byte aligned; its control words must be zero
// (clear_scratch = true issues the memset here; batch callers clear once and pass false).
inline uint32_t *scan_align(uint32_t *p) { return (uint32_t *)(((uintptr_t)p + 7) & ~(uintptr_t)7); }

inline int clear_scan_area(uint32_t *words, size_t n_max, hipStream_t stream) {
  GGMS_HIP(hipMemsetAsync(scan_align(words), 0, scan_clear_words(n_max) * sizeof(uint32_t), stream));
  return GGMS_OK;
}

template <typename ValueF, typename EmitF>
inline int tile_scan(ValueF value, EmitF emit, size_t n_max, Count n, ScanArea area,
                     const uint32_t *base_in, uint32_t *total32_out, uint64_t *total64_out,
                     hipStream_t stream, uint64_t *mirror_a = nullptr, uint64_t *mirror_b = nullptr) {
  if (n_max <= kSmallScan) { // needs no scratch at all
    hipLaunchKernelGGL((k_small_scan<ValueF, EmitF>), dim3(1), dim3(1024), 0, stream, value, emit, n, base_in,
                       total32_out, total64_out, mirror_a, mirror_b);
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  const size_t nt = num_tiles_for(n_max);
  uint32_t *ctl = scan_align(area.words); // 64-bit descriptors need 8-byte alignment
  unsigned long long *desc = reinterpret_cast<unsigned long long *>(ctl + 8);
  const bool single = scan_single_pass(n_max);
  if (!area.cleared && single) { // only the single-pass kernel needs zeroed control words
    int rc = clear_scan_area(area.words, n_max, stream);
    if (rc != GGMS_OK) return rc;
  }
  const int grid = grid_for(nt, 1);
  if (!single) {
    uint32_t *tile_sums = ctl + 8 + 2 * (nt + 1); // behind the descriptors
    uint32_t *tile_prefix = tile_sums + nt + 1;
    hipLaunchKernelGGL((k_tile_reduce<ValueF>), dim3(grid), dim3(kBlock), 0, stream, value, n, tile_sums, area.stash);
    hipLaunchKernelGGL((k_tile_prefix_t<0>), dim3(1), dim3(kBlock), 0, stream, tile_sums, n, tile_prefix, base_in,
                       total32_out, total64_out, mirror_a, mirror_b);
    hipLaunchKernelGGL((k_tile_apply<ValueF, EmitF>), dim3(grid), dim3(kBlock), 0, stream, value, emit, n, tile_prefix,
                       area.stash);
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  hipLaunchKernelGGL((k_tile_scan<ValueF, EmitF>), dim3(grid), dim3(kBlock), 0, stream, value, emit, n, desc,
                     next_scan_epoch(), base_in, total32_out, total64_out, mirror_a, mirror_b, area.status_word(),
                     scan_patience(), scan_delay_word().exchange(0u));
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

// ---- two sums over one input, one pass ---------------------------------------------------------------------------
// Some samplers need two running sums of the same per-seed read (khop0: output offsets = sum of min(deg, fanout), draw
// bases = sum of the neighbour positions beyond fanout).  Two scans are two launches, two passes over the seeds' list
// heads and two look-back chains on a stream whose small kernels are latency chains; this form carries both sums
// through ONE launch: value(i) -> {a, b}, emit(i, a, b, prefix_a, prefix_b).  Channel A's descriptors sit at
// desc[0, tiles), channel B's right behind them at desc[tiles, 2 tiles) (tiles = the launch's upper bound), so a
// launch uses ONE contiguous range of the area.  side.start(n) runs once, on workgroup 0 (a batch prologue, a counter
// reset, ...).  Inputs beyond kSinglePassTiles take two plain scans (tile_scan2 below).
struct U2 {
  uint32_t a, b;
};
template <typename ValueF2, int CH>
struct PickChannel { // by value: it is also handed to kernels (the two-launch fallback)
  ValueF2 v;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
    const U2 x = v(i);
    return CH ? x.b : x.a;
  }
};
struct NoSide {
  __device__ __forceinline__ void start(uint64_t, uint32_t, uint32_t) const {}
};

template <typename ValueF2, typename EmitF2, typename Side>
__global__ __launch_bounds__(kBlock) void k_tile_scan2(ValueF2 value, EmitF2 emit, Side side, Count n_arg,
                                                       unsigned long long *desc, uint32_t b_off, uint32_t epoch,
                                                       uint64_t *total_a_out, uint32_t *err, uint32_t patience) {
  constexpr uint32_t ROUNDS = kTile / kBlock, FLAG_A = 1, FLAG_P = 2;
  __shared__ uint32_t smem[kBlock / kWave];
  __shared__ uint32_t s_prefix[2];
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + kTile - 1) / kTile;
  if (blockIdx.x == 0) side.start(n, threadIdx.x, kBlock);
  unsigned long long *const desc_b = desc + b_off;
  for (uint64_t round = 0;; ++round) {
    const uint64_t tile = (uint64_t)blockIdx.x + round * gridDim.x;
    if (tile >= num_tiles) break;
    U2 v[ROUNDS];
    uint32_t ea[ROUNDS], eb[ROUNDS];
    uint32_t run_a = 0, run_b = 0;
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
      const uint64_t i = tile * kTile + r * kBlock + threadIdx.x;
      v[r] = (i < n) ? value(i) : U2{0u, 0u};
      uint32_t ta, tb;
      ea[r] = run_a + block_exclusive_scan(v[r].a, smem, ta);
      eb[r] = run_b + block_exclusive_scan(v[r].b, smem, tb);
      run_a += ta;
      run_b += tb;
    }
    if (threadIdx.x < kWave) { // wave 0 publishes both sums and looks back for both
      const uint32_t lane = threadIdx.x;
      uint32_t pa = 0, pb = 0;
      if (tile == 0) {
        if (lane == 0) {
          __hip_atomic_store(&desc[0], scan_desc(epoch, FLAG_P, run_a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&desc_b[0], scan_desc(epoch, FLAG_P, run_b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        if (lane == 0) {
          __hip_atomic_store(&desc[tile], scan_desc(epoch, FLAG_A, run_a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&desc_b[tile], scan_desc(epoch, FLAG_A, run_b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const PickChannel<ValueF2, 0> va{value};
        const PickChannel<ValueF2, 1> vb{value};
        pa = scan_lookback(desc, tile, epoch, err, patience, TileSumHelp<PickChannel<ValueF2, 0>>{va, n, 0u});
        pb = scan_lookback(desc_b, tile, epoch, err, patience, TileSumHelp<PickChannel<ValueF2, 1>>{vb, n, 0u});
        if (lane == 0) {
          __hip_atomic_store(&desc[tile], scan_desc(epoch, FLAG_P, pa + run_a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&desc_b[tile], scan_desc(epoch, FLAG_P, pb + run_b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (lane == 0) {
        s_prefix[0] = pa;
        s_prefix[1] = pb;
        if (tile + 1 == num_tiles && total_a_out) *total_a_out = (uint64_t)(pa + run_a);
      }
    }
    __syncthreads();
    const uint32_t pa = s_prefix[0], pb = s_prefix[1];
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
      const uint64_t i = tile * kTile + r * kBlock + threadIdx.x;
      if (i < n) emit(i, v[r].a, v[r].b, pa + ea[r], pb + eb[r]);
    }
    __syncthreads(); // s_prefix is rewritten next iteration
  }
  if (threadIdx.x == 0 && num_tiles == 0 && blockIdx.x == 0 && total_a_out) *total_a_out = 0;
}

template <typename ValueF2, typename EmitF2, typename Side>
__global__ __launch_bounds__(1024) void k_small_scan2(ValueF2 value, EmitF2 emit, Side side, Count n_arg,
                                                      uint64_t *total_a_out) {
  __shared__ uint32_t wsum[2][16];
  const uint64_t n = n_arg.get();
  side.start(n, threadIdx.x, 1024);
  const uint32_t w = threadIdx.x >> 6;
  const uint64_t i = threadIdx.x;
  const U2 v = (i < n) ? value(i) : U2{0u, 0u};
  const uint32_t ia = wave_inclusive_scan(v.a), ib = wave_inclusive_scan(v.b);
  if (lane_id() == 63) {
    wsum[0][w] = ia;
    wsum[1][w] = ib;
  }
  __syncthreads();
  uint32_t ba = 0, bb = 0, ta = 0;
#pragma unroll
  for (uint32_t k = 0; k < 16; ++k) {
    const uint32_t sa = wsum[0][k], sb = wsum[1][k];
    if (k < w) {
      ba += sa;
      bb += sb;
    }
    ta += sa;
  }
  if (i < n) emit(i, v.a, v.b, ba + ia - v.a, bb + ib - v.b);
  if (threadIdx.x == 0 && total_a_out) *total_a_out = (uint64_t)ta;
}

// whether tile_scan2 over n_max items is ONE launch (the caller may then hang a batch prologue on it)
inline bool scan2_single_launch(size_t n_max) { return n_max <= kSmallScan || scan_single_pass(n_max); }
// descriptor words (behind the 8 control words) a one-launch tile_scan2 over n_max items uses
inline size_t scan2_desc_words(size_t n_max) { return n_max <= kSmallScan ? 0 : 4 * num_tiles_for(n_max); }

// emit adapters for the two-launch fallback: the pair's other half is recomputed by the second scan
template <typename EmitF2, int CH>
struct EmitChannel {
  EmitF2 emit;
  __device__ __forceinline__ void operator()(uint64_t i, uint32_t v, uint32_t excl) const { emit.template one<CH>(i, v, excl); }
};

// area: control words + >= 4 (tiles + 1) descriptor words (single pass), `cleared` as for tile_scan.  side must be
// NoSide unless scan2_single_launch(n_max).
template <typename ValueF2, typename EmitF2, typename Side>
inline int tile_scan2(ValueF2 value, EmitF2 emit, Side side, size_t n_max, Count n, ScanArea area, uint64_t *total_a_out,
                      hipStream_t stream) {
  if (n_max <= kSmallScan) {
    hipLaunchKernelGGL((k_small_scan2<ValueF2, EmitF2, Side>), dim3(1), dim3(1024), 0, stream, value, emit, side, n, total_a_out);
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  const size_t nt = num_tiles_for(n_max);
  uint32_t *ctl = scan_align(area.words);
  unsigned long long *desc = reinterpret_cast<unsigned long long *>(ctl + 8);
  if (scan_single_pass(n_max)) {
    if (!area.cleared) GGMS_HIP(hipMemsetAsync(ctl, 0, (8 + 4 * (nt + 1)) * sizeof(uint32_t), stream));
    hipLaunchKernelGGL((k_tile_scan2<ValueF2, EmitF2, Side>), dim3(grid_for(nt, 1)), dim3(kBlock), 0, stream, value, emit,
                       side, n, desc, (uint32_t)nt, next_scan_epoch(), total_a_out, area.status_word(), scan_patience());
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  // large inputs: two plain scans (three launches each; no inter-workgroup wait), each emitting its own half
  int rc = tile_scan(PickChannel<ValueF2, 0>{value}, EmitChannel<EmitF2, 0>{emit}, n_max, n, area, nullptr, nullptr,
                     total_a_out, stream);
  if (rc != GGMS_OK) return rc;
  ScanArea again = area;
  again.cleared = true;
  return tile_scan(PickChannel<ValueF2, 1>{value}, EmitChannel<EmitF2, 1>{emit}, n_max, n, again, nullptr, nullptr,
                   nullptr, stream);
}

} // namespace ggms
