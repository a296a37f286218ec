// sample_khop.hip -- uniform fan-out neighbour samplers (khop3, khop0, khop2).
//
// Reference: GPUSampleKHop3 (cuda/cuda_sampling_khop3.cu:76-146, host :234-318) and
// GPUSampleKHop0 NEW_ALGO (cuda/cuda_sampling_khop0.cu:102-153, host :243-335),
// GPUSampleKHop2 ORIGIN_KHOP2 (cuda/cuda_sampling_khop2.cu:46-95, host :196-262).
//
// What is kept bit-for-bit: which RNG stream serves which seed, the order in
// which a stream's draws are consumed, and the (seed order, slot order) layout
// of the compact COO.  What is redesigned for MI355X:
//   * every seed emits exactly min(deg, fanout) edges, so output offsets are an
//     exclusive scan of that quantity and the sampler writes the compact COO
//     directly: the reference's padded tmp arrays + count/scan/compact passes
//     (khop3.cu:148-230,272-302) and their four host syncs disappear;
//   * khop3: the reference spends 16 lanes of a warp on one seed although its
//     rejection loop is serial by construction (one shared XORWOW state, one
//     insert per draw, khop3.cu:125-131).  Here the 16 lanes of a group take 16
//     consecutive draws of the stream per round and resolve them together (set
//     lookup, in-round first-occurrence dedup, ranking); seeds that take all their
//     neighbours never enter the loop; see k_khop3_positions below;
//   * khop2: one lane per stream as the assignment demands, compact COO written
//     directly at the seed's scanned offset;
//   * khop0: one lane per logical reservoir lane (32 per seed, as the RNG stream
//     assignment demands) only GENERATES the draws; resolving them (modulo, slot
//     update) is a separate, fully parallel kernel.  The racy atomicExch
//     (khop0.cu:144-148) becomes an LDS atomicMax on the candidate position, i.e.
//     highest-j-wins, deterministic.
#include <algorithm>
#include <atomic>

#include "ggms_internal.h"

namespace ggms {

// ---- phase A: per-seed edge count = min(deg, fanout) ------------------------
struct SeedCount {
  GraphView g;
  const uint32_t *input;
  uint32_t fanout;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
    uint32_t len;
    g.neighbours(input[i], len);
    return len < fanout ? len : fanout;
  }
};
struct StoreOffset {
  uint32_t *offset;
  __device__ __forceinline__ void operator()(uint64_t i, uint32_t, uint32_t excl) const { offset[i] = excl; }
};

// ---- khop3 -------------------------------------------------------------------
// Group (b, y) of the reference grid == stream id i = 8 b + y; it serves seeds
// 128 b + y + 8 k, k = 0..15, in that order (khop3.cu:86-89,106).
//
// ONE launch per layer (k_khop3_fused); a workgroup takes reference block b (128 seeds) from a ticket and
//   1. loads the 128 seeds' neighbour-list heads (pointer, degree) into LDS -- the only time they are read;
//   2. scans min(deg, fanout) over the 128 seeds (their slice of the compact COO) and publishes the tile
//      aggregate for the decoupled look-back (tile_scan.h) -- the reference's padded tmp arrays and its
//      count / DeviceScan / compact passes (khop3.cu:148-230,272-302) are this scan;
//   3. draws the positions: the reference's own geometry -- 16 lanes per stream, 8 streams -- but the 16 lanes
//      do useful work.  The stream is serial only in its XORWOW recurrence (9 ALU ops per draw); everything
//      else is done 16 draws at a time: every lane of the group steps the generator 16 times and keeps draw
//      number `lig`, then the 16 candidates are reduced mod deg, looked up in the group's LDS set,
//      de-duplicated against EARLIER candidates with DPP row shifts (first occurrence wins, exactly what the
//      one-draw-at-a-time loop of khop3.cu:125-131 yields), ranked with a ballot, and the accepted ones kept
//      in rank order.  If the set completes at candidate t*, the generator is rewound to just after draw t*,
//      so the stream position is the reference's.  Positions stay in LDS;
//   4. looks back for the tile's base offset (by now the predecessors have published: the wait hides behind 3);
//   5. sweeps the tile's (seed, slot) pairs with all lanes: neighbour load, COO write (src value, neighbour id)
//      and -- INSERT, direct dedup table -- the neighbour's table entry (DedupInsert::enter: one returning
//      atomicMin, which is also all that the owner scan needs to know).  The sampler's ALU-bound phase 3 and the
//      memory-bound phase 5 of different workgroups overlap on the chip instead of being two kernels.
template <int S>
__device__ __forceinline__ uint32_t row_shr(uint32_t v) {
  // lane l of a 16-lane row reads lane l - S of the same row; lanes l < S read 0 (bound_ctrl), which lets the
  // compiler fold the move into the consuming compare (v_cmp_eq_u32_dpp) -- callers shift values that are never 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + S, 0xf, 0xf, true);
}

// Only seeds with more neighbours than `fanout` enter the serial loop; the "take them all" seeds consume no
// draws.  GPW = groups per wave64: 1 gives every group a wave of its own (small frontiers: the chip has idle SIMDs and
// a group does not wait for a neighbour), 2 beyond 256 tiles; the idle lanes of a sparse wave still work in phase 5.
// Registers: the kernel is held at 79 VGPRs (6 waves per SIMD, __launch_bounds__'s second argument; 8 waves would
// spill) and its look-ups into argument arrays (IdxMap, PtrSet) are short runtime loops, not unrolled selects -- the
// 16-way unrolled form kept 34 argument words live and made the kernel spill 122 SGPRs
// (profiles/r04_ab_vs_r03_same_box.txt, r04_ab_khop3_waves_per_simd.txt).
// what tile t's descriptor holds -- the edges its 128 seeds will produce -- computed by one wave from the kernel's input
// (scan_lookback's Help: a look-back that has waited long enough no longer depends on tile t's workgroup running)
struct Khop3TileHelp {
  static constexpr bool kCan = true;
  const GraphView &g;
  const uint32_t *input;
  uint64_t n;
  uint32_t fanout;
  __device__ __forceinline__ uint32_t operator()(uint64_t t) const {
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
      const uint64_t i = 128 * t + 64 * k + (threadIdx.x & 63u);
      if (i < n) {
        uint32_t len;
        (void)g.neighbours(input[i], len);
        acc += len < fanout ? len : fanout;
      }
    }
    return wave_reduce_sum(acc);
  }
};

// FIRST (the first layer of a batch whose seeds are promised distinct; one tile per workgroup, INSERT): the launch is
// also the batch's first kernel -- workgroup 0 runs the batch prologue (it must not touch the tile descriptors this
// launch uses: the host keeps them out of its ranges) and every seed is entered where it is read: head of the unique
// list, table word {pending, position}.  A seed's word is smaller than any neighbour instance's, so it wins whenever it
// arrives; if a neighbour instance of this launch got there first the seed tells it so through `lost`, exactly as any
// smaller arrival does (DedupInsert) -- the owner scan and the end-of-batch look-ups need nothing new.
template <int GPW, bool INSERT, bool FIRST>
__global__ __launch_bounds__(128 * (4 / GPW), 6) void k_khop3_fused(GraphView g, const uint32_t *__restrict__ input,
                                                                 Count n_arg, uint32_t fanout, uint32_t fanout_magic,
                                                                 uint32_t *__restrict__ out_src,
                                                                 uint32_t *__restrict__ out_dst, SrcMode sm,
                                                                 uint32_t *__restrict__ states, uint32_t set_mask,
                                                                 uint32_t multi, FusedScan fs, DedupInsert di,
                                                                 FirstLayer fl) {
  constexpr uint32_t HASH_EMPTY = 0xffffffffu, NT = 128 * (4 / GPW), FLAG_A = 1, FLAG_P = 2;
  extern __shared__ uint32_t s_pos[];  // [128][fanout]: sampled positions of the tile's seeds
  __shared__ uint32_t set_tab[8][128]; // one open-addressing set per group; set_mask + 1 slots in use
  __shared__ const uint32_t *s_ptr[128];
  __shared__ uint32_t s_len[128], s_rid[128], s_off[129];
  __shared__ uint64_t s_tile;
  __shared__ uint32_t s_prefix;
  const uint32_t lane = threadIdx.x & 63u;
  const bool active = GPW == 4 || lane < 16u * GPW; // lanes that belong to a group
  const uint64_t n = n_arg.get();
  if constexpr (FIRST) {
    if (blockIdx.x == 0) fl.pro.run(n, threadIdx.x, NT);
  }
  const uint32_t y = (threadIdx.x >> 6) * GPW + (lane >> 4), lig = lane & 15u;
  const uint32_t grp_shift = lane & ~15u; // first lane of my group inside the wave
  uint32_t *const tab = set_tab[active ? y : 0];
  const uint64_t num_tiles = (n + 127) / 128;
  const uint32_t my_s = y + 8u * lig; // my seed of the tile (as a group lane)

  if (active)
    for (uint32_t s = lig; s <= set_mask; s += 16) tab[s] = HASH_EMPTY;
  bool dirty = false; // group-uniform: the set holds entries

  // A launch with a workgroup per tile (every layer but the largest ones) needs no ticket: tile = workgroup id, one
  // memory round trip less on the tile's latency chain.  Nothing is assumed about when a predecessor's workgroup runs:
  // a look-back that has waited long enough computes the missing aggregates itself (Khop3TileHelp).
  const bool one_tile_each = gridDim.x >= num_tiles; // uniform
  for (uint32_t turn = 0;; ++turn) {
    if (one_tile_each) {
      if (turn != 0) break;
    } else {
      if (threadIdx.x == 0) s_tile = take_ticket(fs.tick);
      __syncthreads();
    }
    const uint64_t b = one_tile_each ? (uint64_t)blockIdx.x : s_tile;
    if (b >= num_tiles) break;
    // ---- 1: lane lig of group y fetches seed k = lig of the group: id -> (list, degree, 2^32/deg)
    uint32_t my_len = 0, my_magic = 0;
    unsigned long long seed_old = 0; // FIRST: what my seed's table word held (looked at after phase 3)
    if (active) {
      const uint64_t my_index = 128 * b + my_s;
      const uint32_t *ptr = nullptr;
      uint32_t rid = 0;
      if (my_index < n) {
        rid = input[my_index];
        if constexpr (FIRST) {
          fl.n2o[my_index] = rid;
          seed_old = atomicMin(di.w + rid, make_w1(di.version, 1u, (uint32_t)my_index));
        }
        ptr = g.neighbours(rid, my_len);
        if (my_len > fanout) my_magic = (uint32_t)(4294967296.0 / (double)my_len); // floor(2^32 / len), exact
      }
      s_ptr[my_s] = ptr;
      s_len[my_s] = my_len;
      s_rid[my_s] = rid;
    }
    __syncthreads();
    // ---- 2: offsets of the 128 seeds inside the tile's slice (wave 0, two seeds per lane) + the tile aggregate
    if (threadIdx.x < kWave) {
      const uint32_t a = min(s_len[2 * lane], fanout), c = min(s_len[2 * lane + 1], fanout);
      const uint32_t incl = wave_inclusive_scan(a + c);
      s_off[2 * lane] = incl - (a + c);
      s_off[2 * lane + 1] = incl - c;
      if (lane == 63) {
        s_off[128] = incl;
        __hip_atomic_store(&fs.desc[b], scan_desc(fs.epoch, b == 0 ? FLAG_P : FLAG_A, incl), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- 3: positions (no barrier needed before it: it reads registers and writes s_pos / the group's set)
    if (active) {
      const uint64_t stream = 8 * b + y;
      const uint32_t my_off = my_s * fanout;
      // seeds of the group that draw, in order k = 0..15
      uint32_t todo = (uint32_t)(__ballot(my_len > fanout) >> grp_shift) & 0xffffu;
      if (todo != 0) { // else the stream is not touched (khop3.cu:111-116 draws nothing)
        Xorwow st;
        st.load(states + 6 * stream);
        bool fetch = true;
        uint32_t len = 0, o = 0, magic = 0, count = 0;
        while (todo != 0) {
          if (fetch) {
            const int src_lane = (int)(grp_shift + (uint32_t)__ffs(todo) - 1u);
            len = __shfl(my_len, src_lane, 64);
            o = __shfl(my_off, src_lane, 64);
            magic = __shfl(my_magic, src_lane, 64);
            count = 0;
            fetch = false;
          }
          // ---- one round: 16 consecutive draws of the stream; lane lig keeps draw lig.
          // w = the raw xorshift word of my draw: the generator's v-array is a sliding window over the
          // sequence (old v0..v4, w of draw 0, w of draw 1, ...), which is what the rewind below reads.
          const Xorwow st0 = st;
          uint32_t x = 0;
#pragma unroll
          for (uint32_t i = 0; i < 16; ++i) {
            const uint32_t xi = st.next();
            x = (lig == i) ? xi : x;
          }
          const uint32_t w = x - (st0.d + (lig + 1) * 362437u); // draw = xorshift word + Weyl counter after lig + 1 steps
          // r = x mod len: q' = mulhi(x, floor(2^32/len)) is q or q - 1 (two fix-ups for safety)
          uint32_t r = x - __umulhi(x, magic) * len;
          r = min(r, r - len);
          r = min(r, r - len);
          // already chosen for this seed?  (nothing is, in a seed's first round)
          bool in_set = false;
          if (count != 0) {
            uint32_t pos = r & set_mask;
            for (;;) {
              const uint32_t cur = tab[pos];
              if (cur == HASH_EMPTY) break;
              if (cur == r) { in_set = true; break; }
              pos = (pos + 1) & set_mask;
            }
          }
          // equal to an EARLIER candidate of this round?  (compared as r + 1, never 0)
          const uint32_t rp = r + 1;
          bool dup = false;
          dup |= row_shr<1>(rp) == rp;
          dup |= row_shr<2>(rp) == rp;
          dup |= row_shr<3>(rp) == rp;
          dup |= row_shr<4>(rp) == rp;
          dup |= row_shr<5>(rp) == rp;
          dup |= row_shr<6>(rp) == rp;
          dup |= row_shr<7>(rp) == rp;
          dup |= row_shr<8>(rp) == rp;
          dup |= row_shr<9>(rp) == rp;
          dup |= row_shr<10>(rp) == rp;
          dup |= row_shr<11>(rp) == rp;
          dup |= row_shr<12>(rp) == rp;
          dup |= row_shr<13>(rp) == rp;
          dup |= row_shr<14>(rp) == rp;
          dup |= row_shr<15>(rp) == rp;
          const bool is_new = !in_set && !dup;
          const uint32_t new_mask = (uint32_t)(__ballot(is_new) >> grp_shift) & 0xffffu;
          const uint32_t rank = __popc(new_mask & ((1u << lig) - 1u));
          const uint32_t total_new = __popc(new_mask);
          const uint32_t need = fanout - count;
          const bool accept = is_new && rank < need;
          if (accept) s_pos[o + count + rank] = r; // insertion order == output order (items[] of khop3.cu:64)
          if (total_new >= need) {
            // The set completed at draw t* = the lane of the need-th new candidate.  State after m = t* + 1
            // draws: v_j = element m + j of the window, d = d0 + m * 362437 -- five lane reads, no replay.
            const uint32_t last_mask = (uint32_t)(__ballot(accept && rank == need - 1) >> grp_shift) & 0xffffu;
            const uint32_t m = (uint32_t)__ffs(last_mask); // t* + 1, in 1..16
            uint32_t v[5];
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j) {
              const uint32_t e = m + j; // window element: 0..4 = old v0..v4, 5 + i = w of draw i
              const uint32_t from_draw = __shfl(w, (int)(grp_shift + (e >= 5 ? e - 5 : 0)), 64);
              uint32_t from_old = st0.v4;
              from_old = (e == 3) ? st0.v3 : from_old;
              from_old = (e == 2) ? st0.v2 : from_old;
              from_old = (e == 1) ? st0.v1 : from_old;
              v[j] = (e >= 5) ? from_draw : from_old; // e >= 1 always
            }
            st.v0 = v[0]; st.v1 = v[1]; st.v2 = v[2]; st.v3 = v[3]; st.v4 = v[4];
            st.d = st0.d + m * 362437u;
            if (dirty) { // next seed starts from an empty set
              __builtin_amdgcn_wave_barrier();
              for (uint32_t s = lig; s <= set_mask; s += 16) tab[s] = HASH_EMPTY;
              __builtin_amdgcn_wave_barrier();
              dirty = false;
            }
            todo &= todo - 1;
            fetch = true;
          } else {
            __builtin_amdgcn_wave_barrier();
            if (accept) {
              uint32_t pos = r & set_mask;
              while (atomicCAS(&tab[pos], HASH_EMPTY, r) != HASH_EMPTY) pos = (pos + 1) & set_mask;
            }
            __builtin_amdgcn_wave_barrier();
            count += total_new;
            dirty = true;
          }
        }
        if (lig == 0) st.store(states + 6 * stream);
      }
    }
    if constexpr (FIRST) { // a neighbour instance of THIS launch held my seed's word: it has just been beaten
      const uint64_t my_index = 128 * b + my_s;
      if (active && my_index < n) {
        const unsigned long long mine = make_w1(di.version, 1u, (uint32_t)my_index);
        const uint32_t idx = (uint32_t)seed_old;
        if (seed_old > mine && (seed_old >> 32) == (mine >> 32) && idx >= di.base) di.lost[idx - di.base] = di.tag;
      }
    }
    // ---- 4: base offset of the tile (wave 0); the last tile knows the total
    if (threadIdx.x < kWave) {
      uint32_t prefix = 0;
      if (b != 0) {
        prefix = scan_lookback(fs.desc, b, fs.epoch, fs.err, fs.patience, Khop3TileHelp{g, input, n, fanout});
        if (lane == 0)
          __hip_atomic_store(&fs.desc[b], scan_desc(fs.epoch, FLAG_P, prefix + s_off[128]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        s_prefix = prefix;
        if (b + 1 == num_tiles) *fs.num_out = (uint64_t)prefix + s_off[128];
      }
    }
    __syncthreads();
    // ---- 5: slot j of seed s -> edge prefix + off[s] + j: its position inside the seed's neighbour list is j
    // for a "take them all" seed and the sampler's pick otherwise; written out as (src value, neighbour id)
    const uint32_t prefix = s_prefix;
    const uint32_t slots = 128u * fanout;
    const auto slot_of = [&](uint32_t t, uint32_t &sd, uint32_t &e, uint32_t &pos) -> bool {
      // s = t / fanout (t < 2^14): mulhi by ceil(2^32 / fanout), one fix-up
      sd = fanout == 1 ? t : __umulhi(t, fanout_magic);
      if (sd * fanout > t) --sd;
      const uint32_t j = t - sd * fanout;
      const uint32_t len = s_len[sd];
      if (j >= (len < fanout ? len : fanout)) return false;
      e = prefix + s_off[sd] + j;
      pos = len <= fanout ? j : s_pos[t];
      return true;
    };
    if (multi) {
      // a frontier that fits the chip at once (every workgroup has ONE tile): the sweep runs at latency, so a lane keeps
      // four slots in flight -- loads together, then the atomics together, then their bookkeeping
      for (uint32_t t0 = threadIdx.x; t0 < slots; t0 += 4 * NT) {
        uint32_t sd[4], e[4], pos[4], nbr[4];
        bool ok[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          const uint32_t t = t0 + u * NT;
          ok[u] = t < slots && slot_of(t, sd[u], e[u], pos[u]);
          nbr[u] = ok[u] ? s_ptr[sd[u]][pos[u]] : 0u;
        }
        unsigned long long old[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          if (ok[u]) {
            out_src[e[u]] = sm.value(s_rid[sd[u]], 128 * b + sd[u]);
            out_dst[e[u]] = nbr[u];
            if (INSERT) old[u] = di.issue(nbr[u], e[u]);
          }
        }
        if (INSERT) {
#pragma unroll
          for (uint32_t u = 0; u < 4; ++u)
            if (ok[u]) di.finish(old[u], e[u]);
        }
      }
    } else {
      // (one slot per lane and round on the large frontiers: four at a time costs registers and measured 5 % SLOWER
      // sampling on papers100M when applied everywhere -- those launches are bounded by requests, not by latency)
      for (uint32_t t = threadIdx.x; t < slots; t += NT) {
        uint32_t sd, e, pos;
        if (!slot_of(t, sd, e, pos)) continue;
        const uint32_t nbr = s_ptr[sd][pos];
        out_src[e] = sm.value(s_rid[sd], 128 * b + sd);
        out_dst[e] = nbr;
        if (INSERT) di.enter(nbr, e);
      }
    }
    __syncthreads(); // LDS is rewritten by the next tile
  }
  if (threadIdx.x == 0 && num_tiles == 0 && blockIdx.x == 0) *fs.num_out = 0;
}

// ---- khop0 (reservoir) -------------------------------------------------------
// Block = 128 threads = 4 logical warps of 32 lanes; thread t: x = t % 32, w = t / 32.
// Stream seed = (b*128 + x*4 + w) + num_input (khop0.cu:114-117); the stream serves the warp's 16 seeds in
// turn, position j >= fanout of a seed being drawn by lane j % 32.
//
// The reference resolves every draw where it is taken (modulo, slot exchange), so a lane that meets a
// 17 000-neighbour node runs ~530 draws x ~60 instructions on its own.  Here the serial part is only the
// generator: k_khop0_generate steps each stream through its draws (9 ALU ops + one store per draw) and parks
// the raw 32-bit draws in a buffer, laid out by (seed, position); k_khop0_resolve then resolves them with one
// wave per seed -- modulo, LDS atomicMax on the slot (highest position wins, the canonical reading of the racy
// atomicExch of khop0.cu:144-148), output.  The buffer holds `cap` draws; a seed whose draws do not fit is
// resolved in place by the generating lanes, the old way, so the capacity only affects speed.
// One pass over the seeds' list heads for both running sums of a khop0 layer (tile_scan2): a = edges the seed emits
// (min(deg, fanout): its slice of the compact COO), b = draws it consumes (one per neighbour position beyond fanout).
struct Khop0Count {
  GraphView g;
  const uint32_t *input;
  uint32_t fanout;
  __device__ __forceinline__ U2 operator()(uint64_t i) const {
    uint32_t len;
    g.neighbours(input[i], len);
    return len > fanout ? U2{fanout, len - fanout} : U2{len, 0u};
  }
};
// ... and what rides on that pass: the two prefix arrays and -- first layer of a batch with distinct seeds, direct
// table -- the seeds themselves (k_seed_enter's work: table word {pending, i}, head of the unique list)
struct Khop0Store {
  uint32_t *offset, *draw_base;
  unsigned long long *w; // NULL: the seeds are not entered here
  uint32_t version;
  const uint32_t *seeds;
  uint32_t *n2o;
  template <int CH>
  __device__ __forceinline__ void one(uint64_t i, uint32_t, uint32_t excl) const {
    (CH ? draw_base : offset)[i] = excl;
  }
  __device__ __forceinline__ void operator()(uint64_t i, uint32_t, uint32_t, uint32_t pa, uint32_t pb) const {
    offset[i] = pa;
    draw_base[i] = pb;
    if (w) {
      const uint32_t key = seeds[i];
      n2o[i] = key;
      atomicMin(w + key, make_w1(version, 1u, (uint32_t)i)); // not returning: nothing of this batch was entered before
    }
  }
};
struct Khop0Side { // workgroup 0 of the pass: the heavy list starts empty; the batch prologue where this is a batch's first kernel
  uint32_t *heavy_count;
  BatchPrologue pro;
  uint32_t has_pro;
  __device__ __forceinline__ void start(uint64_t n, uint32_t tid, uint32_t nthreads) const {
    if (tid == 0) *heavy_count = 0u;
    if (has_pro) pro.run(n, tid, nthreads);
  }
};

// Reservoir slots: LDS up to kKhop0LdsFanout; beyond that (BIG -- the reference has no bound, its slots are its
// padded output array) the seed's slice of out_dst itself holds the winning positions until they are replaced by
// the neighbours found there: the same atomicMax, on HBM, with agent-scope accesses instead of LDS ones.
constexpr uint32_t kKhop0LdsFanout = 2048;
template <bool BIG>
__device__ __forceinline__ void slot_init(uint32_t *p, uint32_t v) {
  if constexpr (BIG) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool BIG>
__device__ __forceinline__ uint32_t slot_read(uint32_t *p) {
  if constexpr (BIG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool BIG>
__device__ __forceinline__ void slot_fence_wave() {
  __builtin_amdgcn_wave_barrier();
  if constexpr (BIG) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
  else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// A seed with more parked draws than this is resolved by a whole workgroup of the resolve kernel, not by one 16-lane group
constexpr uint32_t kKhop0Heavy = 1024;

template <bool BIG>
__global__ __launch_bounds__(128) void k_khop0_generate(GraphView g, const uint32_t *__restrict__ input, Count n_arg,
                                                        uint32_t fanout, const uint32_t *__restrict__ offset,
                                                        const uint32_t *__restrict__ draw_base,
                                                        uint32_t *__restrict__ raw, uint32_t cap,
                                                        uint32_t *__restrict__ out_src,
                                                        uint32_t *__restrict__ out_dst, SrcMode sm,
                                                        uint32_t *__restrict__ heavy_count,
                                                        uint32_t *__restrict__ heavy_list, DedupInsert di) {
  extern __shared__ uint32_t slot_j[]; // [4][fanout]: winning position per reservoir slot (in-place seeds only)
  const uint64_t n = n_arg.get();
  const uint32_t x = threadIdx.x & 31, w = threadIdx.x >> 5;
  const uint32_t half = threadIdx.x & 32u; // first lane of my logical warp inside the wave64
  const uint64_t num_blocks = (n + 63) / 64;
  uint32_t first_j = x;
  while (first_j < fanout) first_j += 32; // first position of this lane that is drawn for
  for (uint64_t b = blockIdx.x; b < num_blocks; b += gridDim.x) {
    Xorwow st;
    st.init((uint64_t)(b * 128 + x * 4 + w) + n);
    // lane k < 16 of the logical warp fetches seed k of the warp (index 64 b + w + 4 k): degree and draw base,
    // so that the serial walk below waits for memory once, not once per seed
    const uint64_t my_index = 64 * b + w + 4 * (uint64_t)(x & 15);
    uint32_t my_len = 0, my_base = 0;
    if (x < 16 && my_index < n) {
      g.neighbours(input[my_index], my_len);
      my_base = draw_base[my_index];
    }
    { // the seeds whose parked draws one 16-lane group should not walk alone: listed here (the resolve kernel's
      // workgroups take them first, a whole workgroup per seed).  One counter update per wave.
      const bool heavy = x < 16 && my_index < n && my_len > fanout && my_len - fanout > kKhop0Heavy &&
                         (uint64_t)my_base + (my_len - fanout) <= cap;
      const uint64_t hm = __ballot(heavy);
      if (hm) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(hm);
        uint32_t at = 0;
        if ((threadIdx.x & 63u) == leader) at = atomicAdd(heavy_count, (uint32_t)__popcll(hm));
        at = __shfl(at, (int)leader, 64);
        if (heavy) heavy_list[at + (uint32_t)__popcll(hm & ((1ull << (threadIdx.x & 63u)) - 1ull))] = (uint32_t)my_index;
      }
    }
    for (uint32_t k = 0; k < 16; ++k) {
      const uint64_t index = 64 * b + w + 4 * (uint64_t)k;
      const uint32_t len = __shfl(my_len, (int)(half + k), 64);
      const uint32_t base = __shfl(my_base, (int)(half + k), 64);
      if (index >= n || len <= fanout) continue; // no draws; k_khop0_resolve copies short lists
      uint32_t j = first_j;
      if ((uint64_t)base + (len - fanout) <= cap) {
#pragma unroll 5 // the XORWOW registers rotate with period 5: no moves in the unrolled body
        for (; j < len; j += 32) raw[base + (j - fanout)] = st.next();
        continue;
      }
      // does not fit the draw buffer: resolve in place (uniform per logical warp)
      const uint32_t rid = input[index];
      uint32_t len2;
      const uint32_t *edges = g.neighbours(rid, len2);
      const uint32_t o = offset[index];
      const uint32_t sv = sm.value(rid, index);
      uint32_t *const my_slots = BIG ? out_dst + o : slot_j + w * fanout;
      for (uint32_t s0 = x; s0 < fanout; s0 += 32) slot_init<BIG>(&my_slots[s0], s0); // slot s starts as position s
      slot_fence_wave<BIG>();
      for (; j < len; j += 32) {
        const uint32_t kk = st.next() % (j + 1);
        if (kk < fanout) atomicMax(&my_slots[kk], j); // highest j wins
      }
      slot_fence_wave<BIG>();
      for (uint32_t s0 = x; s0 < fanout; s0 += 32) {
        const uint32_t nbr = edges[slot_read<BIG>(&my_slots[s0])];
        out_src[o + s0] = sv;
        out_dst[o + s0] = nbr;
        if (di.w) di.enter(nbr, o + s0); // direct dedup table of the batch: entered where it is produced
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// the resolve step proper: lanes `lane`, `lane + stride`, ... of the draws of one seed into its LDS slots
// U independent loads in flight per lane: 8 for a 16-lane group (at most kKhop0Heavy draws: 8 rounds), 32 for the
// workgroup that takes a listed long list alone -- its rounds are latency, and a 300 K-neighbour hub is the kernel's tail
template <uint32_t U = 8>
__device__ __forceinline__ void khop0_resolve_draws(const uint32_t *__restrict__ raw, uint32_t base, uint32_t extra,
                                                    uint32_t fanout, uint32_t lane, uint32_t stride,
                                                    uint32_t *slots) {
  for (uint32_t d0 = 0; d0 < extra; d0 += stride * U) {
    uint32_t xs[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const uint32_t d = d0 + u * stride + lane;
      xs[u] = d < extra ? raw[base + d] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const uint32_t d = d0 + u * stride + lane;
      const uint32_t j = fanout + d;
      const uint32_t kk = xs[u] % (j + 1);
      if (d < extra && kk < fanout) atomicMax(&slots[kk], j); // highest position wins
    }
  }
}

// Resolve: the listed long lists first -- one whole workgroup per seed (k_khop0_generate listed them; the longest
// tasks start first) -- then 16 lanes per seed (4 seeds per wave, 16 per block): copy (deg <= fanout) or resolve the
// parked draws.  The per-seed chain of dependent loads (id -> degree -> offsets -> draws -> neighbours) is what bounds
// this kernel, so several seeds share a wave.
template <bool BIG>
__global__ __launch_bounds__(kBlock) void k_khop0_resolve(GraphView g, const uint32_t *__restrict__ input, Count n_arg,
                                                          uint32_t fanout, const uint32_t *__restrict__ offset,
                                                          const uint32_t *__restrict__ draw_base,
                                                          const uint32_t *__restrict__ raw, uint32_t cap,
                                                          uint32_t *__restrict__ out_src,
                                                          uint32_t *__restrict__ out_dst, SrcMode sm,
                                                          const uint32_t *__restrict__ heavy_count,
                                                          const uint32_t *__restrict__ heavy_list, uint32_t groups,
                                                          DedupInsert di) {
  extern __shared__ uint32_t slot_j[]; // [groups][fanout]; the heavy pass uses the first [fanout]
  constexpr uint32_t G = 16;
  const uint64_t n = n_arg.get();
  {
    const uint32_t num_heavy = *heavy_count; // uniform
    for (uint32_t h = blockIdx.x; h < num_heavy; h += gridDim.x) {
      const uint32_t index = heavy_list[h];
      const uint32_t rid = input[index];
      uint32_t len;
      const uint32_t *edges = g.neighbours(rid, len);
      const uint32_t o = offset[index], base = draw_base[index];
      const uint32_t sv = sm.value(rid, index);
      uint32_t *const slots = BIG ? out_dst + o : slot_j;
      for (uint32_t s0 = threadIdx.x; s0 < fanout; s0 += kBlock) slot_init<BIG>(&slots[s0], s0);
      if constexpr (BIG) __threadfence();
      __syncthreads();
      khop0_resolve_draws<32>(raw, base, len - fanout, fanout, threadIdx.x, kBlock, slots);
      if constexpr (BIG) __threadfence();
      __syncthreads();
      for (uint32_t s0 = threadIdx.x; s0 < fanout; s0 += kBlock) {
        const uint32_t nbr = edges[slot_read<BIG>(&slots[s0])];
        out_src[o + s0] = sv;
        out_dst[o + s0] = nbr;
        if (di.w) di.enter(nbr, o + s0);
      }
      __syncthreads();
    }
  }
  const uint32_t lig = threadIdx.x & (G - 1), grp = threadIdx.x / G;
  if (grp >= groups) return; // wide fanouts: fewer seeds per block, the LDS slots decide
  const uint64_t stride = (uint64_t)gridDim.x * groups;
  for (uint64_t index = (uint64_t)blockIdx.x * groups + grp; index < n; index += stride) {
    const uint32_t rid = input[index];
    uint32_t len;
    const uint32_t *edges = g.neighbours(rid, len);
    const uint32_t o = offset[index];
    const uint32_t sv = sm.value(rid, index);
    if (len <= fanout) {
      for (uint32_t j = lig; j < len; j += G) {
        const uint32_t nbr = edges[j];
        out_src[o + j] = sv;
        out_dst[o + j] = nbr;
        if (di.w) di.enter(nbr, o + j);
      }
      continue;
    }
    const uint32_t base = draw_base[index], extra = len - fanout;
    if ((uint64_t)base + extra > cap) continue; // resolved in place by k_khop0_generate
    if (extra > kKhop0Heavy) continue;          // listed: taken by a whole workgroup above
    uint32_t *const my_slots = BIG ? out_dst + o : slot_j + grp * fanout;
    for (uint32_t s0 = lig; s0 < fanout; s0 += G) slot_init<BIG>(&my_slots[s0], s0);
    slot_fence_wave<BIG>();
    khop0_resolve_draws(raw, base, extra, fanout, lig, G, my_slots);
    slot_fence_wave<BIG>();
    for (uint32_t s0 = lig; s0 < fanout; s0 += G) {
      const uint32_t nbr = edges[slot_read<BIG>(&my_slots[s0])];
      out_src[o + s0] = sv;
      out_dst[o + s0] = nbr;
      if (di.w) di.enter(nbr, o + s0);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- khop2 (in-place partial Fisher-Yates) -----------------------------------
// Thread t of block b owns stream 256 b + t and serves seeds 1024 b + t + 256 r, r = 0..3
// (khop2.cu:53-61).  Draw j of a seed picks position curand % (len - j) of the seed's own
// neighbour list, emits it and swaps it to the shrinking tail (:82-91): `indices` is
// permuted in place and the next batch samples from the permuted lists.  One lane per
// stream, as the stream assignment demands, and a draw costs one round trip to the list
// (it must see the previous swap) -- but only the draws of ONE seed depend on each other.
// The generator is pure ALU, so the lane first positions four copies of its stream where
// the four seeds' draws begin (seed r starts after the draws of the seeds before it: 9 ALU
// ops per skipped draw) and then walks the four lists in lock-step: `fanout` round trips
// per lane instead of 4 x fanout, with identical draws and an identical final state.
// The compact COO is written directly at the seed's scanned offset.  Seeds of one call
// must be distinct (two copies of a seed would race on the same list, as in the reference).
// The batch enters the output into its dedup table with a separate launch: a returning atomic
// in the draw loop sits on the lane's dependency chain (measured 0.45 -> 0.62 ms per products step).
// One wave per workgroup (a quarter of a reference block): every memory instruction of this kernel touches 64
// different lines, so the CU's address path, shared by the waves of a workgroup, is what a 256-thread block waits
// for (measured: 8000 seeds = 8 blocks took 50 us by themselves); spread over four times as many CUs it does not.
__global__ __launch_bounds__(kWave) void k_sample_khop2(const uint32_t *__restrict__ indptr, uint32_t *indices,
                                                         const uint32_t *__restrict__ input, Count n_arg,
                                                         uint32_t fanout, const uint32_t *__restrict__ offset,
                                                         uint32_t *__restrict__ out_src,
                                                         uint32_t *__restrict__ out_dst,
                                                         uint32_t *__restrict__ states, SrcMode sm) {
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + 1023) / 1024;
  for (uint64_t q = blockIdx.x; q < 4 * num_tiles; q += gridDim.x) {
    const uint64_t b = q >> 2;                                  // reference block
    const uint32_t t = (uint32_t)(q & 3u) * kWave + threadIdx.x; // thread of that block
    const uint64_t sid = b * kBlock + t;
    uint32_t off[4], len[4], o[4], sv[4];
    bool draws[4];
    {
      // the four seeds' heads: branch-free (indices clamped into the tile), so the four id -> indptr chains overlap
      uint64_t index[4];
      uint32_t rid[4], end[4];
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        const uint64_t i = b * 1024 + t + (uint64_t)r * kBlock;
        index[r] = i < n ? i : n - 1; // n >= 1 inside a tile
        rid[r] = input[index[r]];
      }
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        off[r] = indptr[rid[r]];
        end[r] = indptr[rid[r] + 1];
        o[r] = offset[index[r]];
        sv[r] = sm.value(rid[r], index[r]);
      }
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        const bool live = b * 1024 + t + (uint64_t)r * kBlock < n;
        len[r] = live ? end[r] - off[r] : 0u;
        draws[r] = len[r] > fanout;
      }
    }
    // short lists are copied whole: every (seed, position) is independent -- 16 loads in flight
    {
      uint32_t longest = 0;
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) longest = (!draws[r] && len[r] > longest) ? len[r] : longest;
      for (uint32_t j0 = 0; j0 < longest; j0 += 4) {
        uint32_t v[4][4];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
          for (uint32_t u = 0; u < 4; ++u)
            if (!draws[r] && j0 + u < len[r]) v[r][u] = indices[off[r] + j0 + u];
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r)
#pragma unroll
          for (uint32_t u = 0; u < 4; ++u)
            if (!draws[r] && j0 + u < len[r]) {
              out_src[o[r] + j0 + u] = sv[r];
              out_dst[o[r] + j0 + u] = v[r][u];
            }
      }
    }
    if (!(draws[0] || draws[1] || draws[2] || draws[3])) continue; // stream untouched
    Xorwow st[4];
    st[0].load(states + 6 * sid);
#pragma unroll
    for (uint32_t r = 1; r < 4; ++r) { // seed r draws after seeds 0..r-1
      st[r] = st[r - 1];
      if (draws[r - 1])
        for (uint32_t j = 0; j < fanout; ++j) (void)st[r].next();
    }
    for (uint32_t j = 0; j < fanout; ++j) {
      uint32_t sel[4], tail[4], picked[4], moved[4];
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        if (!draws[r]) continue;
        sel[r] = st[r].next() % (len[r] - j);
        tail[r] = len[r] - j - 1;
        picked[r] = indices[off[r] + sel[r]];
        moved[r] = indices[off[r] + tail[r]];
      }
#pragma unroll
      for (uint32_t r = 0; r < 4; ++r) {
        if (!draws[r]) continue;
        out_src[o[r] + j] = sv[r];
        out_dst[o[r] + j] = picked[r];
        indices[off[r] + sel[r]] = moved[r];
        indices[off[r] + tail[r]] = picked[r];
      }
    }
    st[3].store(states + 6 * sid); // past its own draws, or -- a short last list -- still where seed 2 ended
  }
}

// groups per wave of the khop3 kernel: sparse waves while the frontier leaves SIMDs idle (1024 SIMDs; a block
// is 8 groups): one group per wave (512 threads per tile) for the first layers of a batch, two (256 threads) beyond.
// Four groups per wave (128 threads per tile: the densest draws phase) measured 5 % slower on products [25,10], 14 % on
// papers100M [25,10] and 1 % on the papers100M GCN layers: the sweep of a tile is a latency chain, twice the lanes
// halve its rounds (profiles/r03_ab_khop3_groups_per_wave.txt) -- not built any more.
static int khop3_groups_per_wave(size_t blocks) { return blocks * 8 <= 2048 ? 1 : 2; }

size_t sample_ws_words(size_t num_input) {
  return num_input + tile_scan_words(num_input) + 16 + 2 * (num_input / 128 + 2) + kTicketWords + 16;
}

template <int GPW, bool INSERT, bool FIRST>
static int launch_khop3_fused(int grid, size_t lds, hipStream_t s, GraphView g, const uint32_t *input, Count n,
                               uint32_t fanout, uint32_t *out_src, uint32_t *out_dst, SrcMode sm, uint32_t *states,
                               uint32_t set_mask, uint32_t multi, FusedScan fs, DedupInsert di, FirstLayer fl) {
  const uint32_t fanout_magic = (uint32_t)((0x100000000ull + fanout - 1) / fanout); // ceil(2^32 / fanout)
  if (lds > (48u << 10)) { // large fan-outs: more dynamic LDS than the default per-kernel limit (gfx950 has 160 KB)
    static const int raised = raise_dynamic_lds(reinterpret_cast<const void *>(&k_khop3_fused<GPW, INSERT, FIRST>),
                                                128 * 127 * 4, "k_khop3_fused (fanout >= 96)");
    if (raised != GGMS_OK) return raised;
  }
  hipLaunchKernelGGL((k_khop3_fused<GPW, INSERT, FIRST>), dim3(grid), dim3(128 * (4 / GPW)), lds, s, g, input, n, fanout,
                     fanout_magic, out_src, out_dst, sm, states, set_mask, multi, fs, di, fl);
  return GGMS_OK;
}

bool khop3_can_fuse_seeds(size_t num_seeds) { return (num_seeds + 127) / 128 <= grid_cap(); }

// the whole layer in one launch (see k_khop3_fused).  shared_scan: the batch's scan area (cleared by the batch
// prologue); else the workspace holds a private one that is cleared here.  insert (direct dedup table): the
// neighbours are entered into the table on the way out.
int sample_khop3_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t fanout, uint32_t *out_src,
                      uint32_t *out_dst, uint64_t *num_out_dev, uint32_t *states, uint32_t *workspace,
                      const uint32_t *seed_local, int src_local, hipStream_t s, ScanArea *shared_scan,
                      const DedupInsert *insert, const FirstLayer *first) {
  const size_t tiles = (n_max + 127) / 128;
  if (first && (!insert || !shared_scan || !shared_scan->cleared || !khop3_can_fuse_seeds(n_max))) {
    set_error("sample_khop3: the fused first layer needs the batch's dedup insert, its scan area and one tile per workgroup");
    return GGMS_ERR_INVALID;
  }
  uint32_t *ctl = scan_align(shared_scan ? shared_scan->words : workspace);
  // tickets: the batch's next set (zeroed by the batch prologue); a private area keeps its set behind the descriptors
  uint32_t *tick = shared_scan ? take_ticket_set(shared_scan) : ctl + 8 + 2 * (tiles + 1) + 2;
  if (!tick) {
    set_error("sample_khop3: the shared scan area has no ticket set left");
    return GGMS_ERR_INVALID;
  }
  if (!shared_scan)
    GGMS_HIP(hipMemsetAsync(ctl, 0, (8 + 2 * (tiles + 1) + 2 + kTicketWords) * sizeof(uint32_t), s));
  else if (!shared_scan->cleared)
    GGMS_HIP(hipMemsetAsync(ctl, 0, (8 + 2 * (tiles + 1)) * sizeof(uint32_t), s));
  const FusedScan fs{tick, reinterpret_cast<unsigned long long *>(ctl + 8), next_scan_epoch(), num_out_dev,
                     shared_scan ? shared_scan->status_word() : device_status_word(), scan_patience()};
  const SrcMode sm{seed_local, src_local};
  // 64 slots up to fanout 31 (load < 0.5), else the reference's 128 (HASHTABLE_SIZE, khop3.cu:43)
  const uint32_t set_mask = fanout < 32 ? 63u : 127u;
  const int gpw = khop3_groups_per_wave(tiles);
  const int grid = grid_for(tiles, 1);
  // every workgroup has at most one tile: the sweep keeps four slots per lane in flight
  const uint32_t multi = tiles <= grid_cap() ? 1u : 0u;
  const size_t lds = 128 * (size_t)fanout * sizeof(uint32_t);
  const DedupInsert none{};
  const FirstLayer nofl{};
  int rc_l = GGMS_OK;
  if (first) {
    if (gpw == 1) rc_l = launch_khop3_fused<1, true, true>(grid, lds, s, g, input, n, fanout, out_src, out_dst, sm, states, set_mask, multi, fs, *insert, *first);
    else rc_l = launch_khop3_fused<2, true, true>(grid, lds, s, g, input, n, fanout, out_src, out_dst, sm, states, set_mask, multi, fs, *insert, *first);
  } else if (insert) {
    if (gpw == 1) rc_l = launch_khop3_fused<1, true, false>(grid, lds, s, g, input, n, fanout, out_src, out_dst, sm, states, set_mask, multi, fs, *insert, nofl);
    else rc_l = launch_khop3_fused<2, true, false>(grid, lds, s, g, input, n, fanout, out_src, out_dst, sm, states, set_mask, multi, fs, *insert, nofl);
  } else {
    if (gpw == 1) rc_l = launch_khop3_fused<1, false, false>(grid, lds, s, g, input, n, fanout, out_src, out_dst, sm, states, set_mask, multi, fs, none, nofl);
    else rc_l = launch_khop3_fused<2, false, false>(grid, lds, s, g, input, n, fanout, out_src, out_dst, sm, states, set_mask, multi, fs, none, nofl);
  }
  if (rc_l != GGMS_OK) return rc_l;
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

// draws parked per launch: 8 per output slot (a frontier whose mean degree stays under 9 x fanout fits entirely)
size_t khop0_draw_cap(size_t num_input, size_t fanout) {
  const unsigned long long want = 8ull * num_input * fanout;
  return (size_t)(want < 0x7fffffffull ? want : 0x7fffffffull);
}
bool khop0_can_enter_seeds(size_t num_seeds) { return scan2_single_launch(num_seeds); }
size_t khop0_plan_desc_words(size_t num_seeds) { return scan2_desc_words(num_seeds); }
size_t khop0_ws_words(size_t num_input, size_t fanout) {
  return 3 * num_input + tile_scan_words(num_input) + 48 + khop0_draw_cap(num_input, fanout);
}

// One layer = THREE launches: plan (both running sums in one pass, tile_scan2; the batch prologue and the distinct
// seeds' table entries ride on it where this is a batch's first kernel), generate (draws parked, long lists listed),
// resolve (listed lists by whole workgroups first, then 16 lanes per seed).  Rounds 1-4 took six (two scans, generate,
// resolve, a 256 x 1024-thread launch for the long lists, + a seed-entry launch per batch): on a stream that runs
// beside another batch's sampler and a gather every small launch is a chain of loaded round trips
// (profiles/r05_khop0_chain.txt).
int sample_khop0_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t fanout, uint32_t *out_src,
                      uint32_t *out_dst, uint64_t *num_out_dev, uint32_t *workspace, const uint32_t *seed_local,
                      int src_local, hipStream_t s, ScanArea *shared_scan, const DedupInsert *insert,
                      const SeedEnter *enter) {
  const DedupInsert di = insert ? *insert : DedupInsert{}; // w == NULL: no table to enter the output into
  uint32_t *offset = workspace;
  uint32_t *draw_base = offset + n_max;
  uint32_t *heavy_list = draw_base + n_max;
  uint32_t *heavy_count = heavy_list + n_max;
  uint32_t *scan_scr = heavy_count + 16;
  uint32_t *raw = scan_scr + tile_scan_words(n_max) + 16;
  // ggms_debug_set_knob(GGMS_DEBUG_KHOP0_DRAW_CAP) (tests): a SMALLER buffer than the workspace holds -- seeds whose
  // draws do not fit are resolved in place by the generating lanes
  const long long forced_cap = debug_knob(GGMS_DEBUG_KHOP0_DRAW_CAP);
  const size_t full_cap = khop0_draw_cap(n_max, fanout);
  const uint32_t cap = (uint32_t)(forced_cap >= 0 && (size_t)forced_cap < full_cap ? (size_t)forced_cap : full_cap);
  const ScanArea sa = shared_scan ? *shared_scan : ScanArea{scan_scr, false};
  const bool one_launch = scan2_single_launch(n_max);
  if (enter && !one_launch) {
    set_error("sample_khop0: seeds can only be entered by a one-launch plan pass");
    return GGMS_ERR_INVALID;
  }
  Khop0Store store{offset, draw_base, nullptr, 0u, nullptr, nullptr};
  Khop0Side side{heavy_count, BatchPrologue{nullptr, 0, nullptr, 0, nullptr, nullptr}, 0u};
  if (enter) {
    store.w = enter->w;
    store.version = enter->version;
    store.seeds = input;
    store.n2o = enter->n2o;
    side.pro = enter->pro;
    side.has_pro = 1u;
  }
  int rc;
  if (one_launch) {
    rc = tile_scan2(Khop0Count{g, input, fanout}, store, side, n_max, n, sa, num_out_dev, s);
  } else { // beyond kSinglePassTiles tiles: two plain scans, nobody to run `side`
    GGMS_HIP(hipMemsetAsync(heavy_count, 0, sizeof(uint32_t), s));
    rc = tile_scan2(Khop0Count{g, input, fanout}, store, NoSide{}, n_max, n, sa, num_out_dev, s);
  }
  if (rc != GGMS_OK) return rc;
  const SrcMode sm{seed_local, src_local};
  const bool big = fanout > kKhop0LdsFanout; // slots in the output array instead of LDS
  const size_t lds = big ? 0 : 4 * fanout * sizeof(uint32_t);
  const int grid_gen = grid_for((n_max + 63) / 64, 1);
  if (big)
    hipLaunchKernelGGL(k_khop0_generate<true>, dim3(grid_gen), dim3(128), lds, s, g, input, n, fanout, offset, draw_base,
                       raw, cap, out_src, out_dst, sm, heavy_count, heavy_list, di);
  else
    hipLaunchKernelGGL(k_khop0_generate<false>, dim3(grid_gen), dim3(128), lds, s, g, input, n, fanout, offset,
                       draw_base, raw, cap, out_src, out_dst, sm, heavy_count, heavy_list, di);
  GGMS_LAUNCH_CHECK();
  // seeds per 256-thread block of the resolve kernel: 16 lanes each, as many as 48 KB of LDS slots allow
  const uint32_t groups = big ? 16u : std::max<uint32_t>(1, std::min<uint32_t>(16, (48u << 10) / (4u * fanout)));
  if (big)
    hipLaunchKernelGGL(k_khop0_resolve<true>, dim3(grid_for(n_max, groups)), dim3(kBlock), 0, s, g, input, n, fanout,
                       offset, draw_base, raw, cap, out_src, out_dst, sm, heavy_count, heavy_list, groups, di);
  else
    hipLaunchKernelGGL(k_khop0_resolve<false>, dim3(grid_for(n_max, groups)), dim3(kBlock),
                       groups * fanout * sizeof(uint32_t), s, g, input, n, fanout, offset, draw_base, raw, cap, out_src,
                       out_dst, sm, heavy_count, heavy_list, groups, di);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

int sample_khop2_impl(const uint32_t *indptr, uint32_t *indices, size_t num_node, const uint32_t *input, size_t n_max,
                      Count n, uint32_t fanout, uint32_t *out_src, uint32_t *out_dst, uint64_t *num_out_dev,
                      uint32_t *states, uint32_t *workspace, const uint32_t *seed_local, int src_local, hipStream_t s,
                      ScanArea *shared_scan) {
  (void)num_node;
  uint32_t *offset = workspace;
  GraphView g{};
  g.indptr = indptr;
  g.indices = indices;
  const ScanArea sa = shared_scan ? *shared_scan : ScanArea{offset + n_max, false};
  int rc = tile_scan(SeedCount{g, input, fanout}, StoreOffset{offset}, n_max, n, sa, nullptr, nullptr,
                     num_out_dev, s);
  if (rc != GGMS_OK) return rc;
  hipLaunchKernelGGL(k_sample_khop2, dim3(grid_for(4 * ((n_max + 1023) / 1024), 1)), dim3(kWave), 0, s, indptr, indices,
                     input, n, fanout, offset, out_src, out_dst, states, SrcMode{seed_local, src_local});
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // namespace ggms

using namespace ggms;

extern "C" {

size_t ggms_sample_workspace_bytes(int sample_type, size_t num_input, size_t fanout) {
  if (sample_type == GGMS_KHOP0) return khop0_ws_words(num_input, fanout) * sizeof(uint32_t);
  return sample_ws_words(num_input) * sizeof(uint32_t);
}

int ggms_sample_khop3(const ggms_graph_t *graph, const ggms_id_t *input, size_t num_input, size_t fanout,
                      ggms_id_t *out_src, ggms_id_t *out_dst, uint64_t *num_out_dev, void *states,
                      size_t num_states, void *workspace, size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev);
  GGMS_CHECK_ARG(fanout > 0 && fanout < 128); // khop3.cu:85
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(input && out_src && out_dst && states && workspace);
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_workspace_bytes(GGMS_KHOP3, num_input, fanout));
  GGMS_CHECK_ARG((uint64_t)num_input * fanout < (1ull << 32));
  GGMS_CHECK_ARG((num_input + 127) / 128 * 8 <= num_states); // assert(i < num_random_states), khop3.cu:89
  GraphView gv;
  if (!view_of(graph, gv)) return GGMS_ERR_INVALID;
  return sample_khop3_impl(gv, input, num_input, count_of(num_input), (uint32_t)fanout, out_src, out_dst,
                           num_out_dev, (uint32_t *)states, (uint32_t *)workspace, nullptr, 0, s);
}

int ggms_sample_khop0(const ggms_graph_t *graph, const ggms_id_t *input, size_t num_input, size_t fanout,
                      ggms_id_t *out_src, ggms_id_t *out_dst, uint64_t *num_out_dev, void *workspace,
                      size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev);
  GGMS_CHECK_ARG(fanout > 0);
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(input && out_src && out_dst && workspace);
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_workspace_bytes(GGMS_KHOP0, num_input, fanout));
  GGMS_CHECK_ARG((uint64_t)num_input * fanout < (1ull << 32));
  GraphView gv;
  if (!view_of(graph, gv)) return GGMS_ERR_INVALID;
  return sample_khop0_impl(gv, input, num_input, count_of(num_input), (uint32_t)fanout, out_src, out_dst,
                           num_out_dev, (uint32_t *)workspace, nullptr, 0, s);
}

int ggms_sample_khop2(const ggms_graph_t *graph, const ggms_id_t *input, size_t num_input, size_t fanout,
                      ggms_id_t *out_src, ggms_id_t *out_dst, uint64_t *num_out_dev, void *states,
                      size_t num_states, void *workspace, size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev);
  GGMS_CHECK_ARG(fanout > 0);
  GGMS_CHECK_ARG(graph->num_part == 0); // CHECK(use_dist_graph == false), dist_loops.cc:219
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(input && out_src && out_dst && states && workspace && graph->indptr && graph->indices);
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_workspace_bytes(GGMS_KHOP2, num_input, fanout));
  GGMS_CHECK_ARG((uint64_t)num_input * fanout < (1ull << 32));
  GGMS_CHECK_ARG((num_input + 1023) / 1024 * 256 <= num_states); // assert(i < num_random_states), khop2.cu:57
  return sample_khop2_impl(graph->indptr, const_cast<uint32_t *>(graph->indices), graph->num_node, input, num_input,
                           count_of(num_input), (uint32_t)fanout, out_src, out_dst, num_out_dev, (uint32_t *)states,
                           (uint32_t *)workspace, nullptr, 0, s);
}

} // extern "C"
