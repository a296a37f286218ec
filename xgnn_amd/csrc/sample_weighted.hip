// sample_weighted.hip -- weighted fan-out samplers (alias method, prefix sums, alias + per-seed dedup)
// and the uniform with-replacement sampler (khop1).
//
// Reference: GPUSampleWeightedKHop, cuda/cuda_sampling_weighted_khop.cu:132-238:
//   sample_weighted_khop :41-76   task t -> seed t / fanout; k = curand % deg; r = curand_uniform;
//                                 dst = r < prob[off+k] ? indices[off+k] : alias[off+k]
//                                 (grid-stride over <= 512 K threads, one stored XORWOW state per thread)
//   cub SortPairs(key = src) :172-181, count_edge / compact_edge :78-128 (drop an entry equal to its successor)
//
// GPUSampleKHop1, cuda/cuda_sampling_khop1.cu:130-236, is the same pipeline with one draw per task
// (dst = indices[off + curand % deg], :65-67) and kKHop1MaxThreads = 512 K (constant.h:71): k_weighted_draw<false>.
//
// GPUSampleWeightedKHopPrefix, cuda/cuda_sampling_weighted_khop_prefix.cu:145-246, again the same pipeline; the
// draw is one curand_uniform scaled by the list's total weight and a binary search in its prefix sums (:59-88):
// k_weighted_draw<2>.
// GPUSampleWeightedKHopHashDedup, cuda/cuda_sampling_weighted_khop_hash_dedup.cu:196-283, is a different
// animal -- the khop2 thread<->stream map, rejection until `fanout` distinct ids: k_weighted_hash_dedup below.
//
// Kept bit-for-bit: task -> thread -> RNG stream assignment (span = ceil(min(tasks, 512K) / 256) * 256),
// two draws per task, the stable order by src, the adjacent-duplicate rule (dedup is partial by design).
// Redesigned: the stable sort of n*fanout (src, dst) pairs by src is a stable sort of the n SEED POSITIONS
// by their id (the fanout tasks of a position are contiguous and stay in order), i.e. fanout-times fewer
// elements through the radix sort; the sorted, expanded stream is never materialised -- the compaction
// pass reads it through the permutation.
#include "ggms_internal.h"
#include "radix_sort.h"

namespace ggms {

constexpr uint64_t kWeightedMaxThreads = 512 * 1024; // Constant::kWeightedKHopMaxThreads, constant.h:72

// per-seed edge count min(deg, fanout) and its scanned offset (hash-dedup sampler: compact COO written directly)
struct MinDegFanout {
  const uint32_t *indptr, *input;
  uint32_t fanout;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
    const uint32_t rid = input[i];
    const uint32_t len = indptr[rid + 1] - indptr[rid];
    return len < fanout ? len : fanout;
  }
};
struct StoreWord {
  uint32_t *out;
  __device__ __forceinline__ void operator()(uint64_t i, uint32_t, uint32_t excl) const { out[i] = excl; }
};

// MODE 0: uniform (khop1), 1: alias method, 2: prefix sums (`prob` = inclusive prefix sums per list)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_weighted_draw(const uint32_t *__restrict__ indptr,
                                                          const uint32_t *__restrict__ indices,
                                                          const float *__restrict__ prob,
                                                          const uint32_t *__restrict__ alias,
                                                          const uint32_t *__restrict__ input, Count n_arg,
                                                          uint32_t fanout, uint32_t *__restrict__ tmp_dst,
                                                          uint32_t *__restrict__ states) {
  const uint64_t n = n_arg.get();
  const uint64_t num_task = n * fanout;
  const uint64_t threads = num_task < kWeightedMaxThreads ? num_task : kWeightedMaxThreads;
  const uint64_t span = (threads + 255) / 256 * 256; // blockDim.x * gridDim.x of the reference launch (:156-160)
  const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (tid >= span || tid >= num_task) return;
  Xorwow st;
  st.load(states + 6 * tid);
  for (uint64_t task = tid; task < num_task; task += span) {
    const uint32_t rid = input[task / fanout];
    const uint32_t off = indptr[rid];
    const uint32_t len = indptr[rid + 1] - off;
    if (len == 0) continue;
    if (MODE == 2) {
      const float x = st.uniform() * prob[off + len - 1]; // prefix.cu:59,66
      if (x <= prob[off]) {
        tmp_dst[task] = indices[off];
      } else {
        uint32_t lo = off, hi = off + len - 1;
        while (hi - lo >= 2) {
          const uint32_t mid = (uint32_t)(((uint64_t)lo + hi) >> 1);
          if (prob[mid] >= x) hi = mid; else lo = mid;
        }
        tmp_dst[task] = indices[hi];
      }
    } else {
      const uint32_t k = st.next() % len;
      if (MODE == 1) {
        const float r = st.uniform();
        tmp_dst[task] = (r < prob[off + k]) ? indices[off + k] : alias[off + k];
      } else {
        tmp_dst[task] = indices[off + k];
      }
    }
  }
  st.store(states + 6 * tid);
}

// sort key of a seed position: its id, or kEmptyKey when it has no neighbour (tmp_src of :58-60)
__global__ __launch_bounds__(kBlock) void k_weighted_keys(const uint32_t *__restrict__ indptr,
                                                          const uint32_t *__restrict__ input, Count n_arg,
                                                          uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint64_t n = n_arg.get();
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const uint32_t rid = input[i];
    keys[i] = (indptr[rid + 1] == indptr[rid]) ? kEmptyKey : rid;
    vals[i] = (uint32_t)i;
  }
}

// element q of the sorted, expanded stream: rank s = q / fanout, slot j = q % fanout
struct SortedStream {
  const uint32_t *keys, *order, *tmp_dst;
  uint32_t fanout;
  Count n;
  __device__ __forceinline__ void at(uint64_t q, uint32_t &src, uint32_t &dst, uint32_t &pos) const {
    const uint64_t s = q / fanout, j = q - s * fanout;
    src = keys[s];
    pos = order[s];
    dst = tmp_dst[(uint64_t)pos * fanout + j];
  }
};
struct KeepFlag { // count_edge, :78-96
  SortedStream ss;
  __device__ __forceinline__ uint32_t operator()(uint64_t q) const {
    const uint64_t total = ss.n.get() * ss.fanout;
    if (q >= total) return 0u; // the scan runs over the upper bound n_max * fanout
    uint32_t src, dst, pos;
    ss.at(q, src, dst, pos);
    if (src == kEmptyKey) return 0u;
    if (q + 1 >= total) return 1u;
    uint32_t nsrc, ndst, npos;
    ss.at(q + 1, nsrc, ndst, npos);
    return (src != nsrc || dst != ndst) ? 1u : 0u;
  }
};
struct KeepEmit { // compact_edge, :98-128
  SortedStream ss;
  uint32_t *out_src, *out_dst;
  const uint32_t *seed_local;
  int src_local;
  DedupInsert di; // w != NULL: the batch's direct dedup table, entered where the edge is written
  __device__ __forceinline__ void operator()(uint64_t q, uint32_t keep, uint32_t at) const {
    if (!keep) return;
    uint32_t src, dst, pos;
    ss.at(q, src, dst, pos);
    out_src[at] = src_local ? (seed_local ? seed_local[pos] : pos) : src;
    out_dst[at] = dst;
    if (di.w) di.enter(dst, at);
  }
};

size_t weighted_ws_words(size_t num_input, size_t fanout) {
  return num_input * fanout + 4 * num_input + sort_scratch_words(num_input) + tile_scan_words(num_input * fanout) + 64;
}

int sample_weighted_impl(const uint32_t *indptr, const uint32_t *indices, const float *prob, const uint32_t *alias,
                         const uint32_t *input, size_t n_max, Count n, uint32_t fanout, uint32_t *out_src,
                         uint32_t *out_dst, uint64_t *num_out_dev, uint32_t *states, uint32_t *workspace,
                         const uint32_t *seed_local, int src_local, hipStream_t s, ScanArea *shared_scan,
                         uint32_t num_node, const DedupInsert *insert) {
  uint32_t *w = workspace;
  uint32_t *tmp_dst = w;  w += n_max * fanout;
  uint32_t *k0 = w;       w += n_max;
  uint32_t *v0 = w;       w += n_max;
  uint32_t *k1 = w;       w += n_max;
  uint32_t *v1 = w;       w += n_max;
  uint32_t *sort_scr = w; w += sort_scratch_words(n_max);
  uint32_t *scan_scr = w;
  const size_t task_max = n_max * fanout;
  const size_t threads = task_max < kWeightedMaxThreads ? task_max : (size_t)kWeightedMaxThreads;
  const dim3 draw_grid((unsigned)((threads + kBlock - 1) / kBlock));
  if (prob && alias)
    hipLaunchKernelGGL(k_weighted_draw<1>, draw_grid, dim3(kBlock), 0, s, indptr, indices, prob, alias, input, n, fanout,
                       tmp_dst, states);
  else if (prob) // prefix sums
    hipLaunchKernelGGL(k_weighted_draw<2>, draw_grid, dim3(kBlock), 0, s, indptr, indices, prob, alias, input, n, fanout,
                       tmp_dst, states);
  else // khop1: uniform with replacement
    hipLaunchKernelGGL(k_weighted_draw<0>, draw_grid, dim3(kBlock), 0, s, indptr, indices, prob, alias, input, n, fanout,
                       tmp_dst, states);
  GGMS_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_weighted_keys, dim3(grid_for(n_max, kBlock)), dim3(kBlock), 0, s, indptr, input, n, k0, v0);
  GGMS_LAUNCH_CHECK();
  // keys are node ids (< num_node) or kEmptyKey: the sort only covers the bits an id can have set, and the empty
  // key (all ones) still sorts behind every id (radix_plan)
  bool in_second = false;
  int rc = radix_sort_pairs(k0, v0, k1, v1, n_max, n, sort_scr, s, num_node, &in_second);
  if (rc != GGMS_OK) return rc;
  const SortedStream ss{in_second ? k1 : k0, in_second ? v1 : v0, tmp_dst, fanout, n};
  // element count of the compaction = n * fanout with n possibly on the device: a Count cannot multiply,
  // so KeepFlag bounds itself by ss.n and the scan runs over the upper bound
  const ScanArea sa = shared_scan ? *shared_scan : ScanArea{scan_scr, false};
  return tile_scan(KeepFlag{ss}, KeepEmit{ss, out_src, out_dst, seed_local, src_local, insert ? *insert : DedupInsert{}}, task_max,
                   count_of(task_max), sa, nullptr, nullptr, num_out_dev, s);
}

// ---- weighted_khop_hash_dedup --------------------------------------------------------------------------
// Thread t of block b owns stream 256 b + t and seeds 1024 b + t + 256 r (hash_dedup.cu:69-80); a seed with more than
// `fanout` neighbours draws alias-method candidates (two draws each) until `fanout` DISTINCT ids are found.  The
// reference keeps the stream's 50-slot table {value, round id} in local memory, keyed by round id = the seed's id and
// therefore never cleared; the table lives in LDS here and is kept exactly as the reference would leave it (an
// earlier seed of the stream with the same id still counts).
//
// The stream <-> seed map and the order of a stream's draws are the reference's; how the draws are RESOLVED is not.
// A try consumes exactly two draws whatever its outcome, so the stream's next tries are known in advance.  One lane
// per stream (round 2) had the slowest list of the launch set its time: a 26-neighbour list asked for 25 distinct
// picks takes hundreds of tries, eight per memory round trip -- 885 us per products step.  Now the G lanes of a group
// serve ONE stream: every lane steps the generator through the round's G R tries (9 ALU ops per draw) and keeps tries
// lig, lig + G, ...; the 3 x G R loads (neighbour, acceptance probability, alias) of the round are in flight
// together; a read-only probe of the table then drops the tries whose candidate is already there (most of
// them, in a long rejection run -- a present entry stays present, entries are only ever added), and the rest go
// through the reference's insert one after the other in try order, so that accepted picks, their order and the
// table's layout are exactly the sequential ones.  If the seed completes at try t*, the generator is put back to just
// after it.
constexpr uint32_t kDedupSlots = 50;       // hash_dedup.cu:42
constexpr uint32_t kDedupMaxTries = 65536; // the reference spins forever on a list without `fanout` distinct ids
constexpr uint32_t kDedupMaxProbes = 64;   // ... and on a full table (a seed id met twice by one stream, fanout >= 25)

template <uint32_t G, uint32_t R> // lanes per stream (a power of two <= 16), tries per lane and round
__global__ __launch_bounds__(kWave) void k_weighted_hash_dedup(const uint32_t *__restrict__ indptr,
                                                                const uint32_t *__restrict__ indices,
                                                                const float *__restrict__ prob,
                                                                const uint32_t *__restrict__ alias,
                                                                const uint32_t *__restrict__ input, Count n_arg,
                                                                uint32_t fanout, const uint32_t *__restrict__ offset,
                                                                uint32_t *__restrict__ out_src,
                                                                uint32_t *__restrict__ out_dst,
                                                                uint32_t *__restrict__ states, SrcMode sm,
                                                                DedupInsert di) {
  constexpr uint32_t SPW = kWave / G; // streams per wave
  __shared__ uint32_t val[SPW][kDedupSlots], round_of[SPW][kDedupSlots]; // one table per group = per stream
  __shared__ uint32_t picked[SPW][kDedupSlots];                          // the current seed's accepted picks, in order
  const uint64_t n = n_arg.get();
  const uint32_t lane = threadIdx.x, g = lane / G, lig = lane % G, grp_shift = lane & ~(G - 1u);
  constexpr uint32_t GMASK = G >= 32 ? 0xffffffffu : (1u << (G & 31u)) - 1u, WPB = kBlock / SPW; // waves per reference block of 256 streams
  uint32_t *const tv = val[g], *const tr = round_of[g];
  const uint64_t num_tiles = (n + 1023) / 1024;
  for (uint64_t q = blockIdx.x; q < WPB * num_tiles; q += gridDim.x) { // WPB waves x SPW streams = one reference block
    const uint64_t b = q / WPB;                            // reference block
    const uint32_t tb = (uint32_t)(q % WPB) * SPW + g;     // thread of that block = the stream this group serves
    for (uint32_t z = lig; z < kDedupSlots; z += G) { tv[z] = kEmptyKey; tr[z] = kEmptyKey; } // :72-76
    __builtin_amdgcn_wave_barrier();
    const uint64_t sid = b * kBlock + tb;
    Xorwow st;
    st.load(states + 6 * sid);
    bool drew = false;
    // the stream's four seeds are known up front: their ids, list bounds and output offsets are fetched together (the
    // seeds themselves must be served one after the other: the generator's state after a seed depends on its tries)
    uint32_t rid_[4], off_[4], len_[4], o_[4], sv_[4];
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
      const uint64_t index = b * 1024 + tb + (uint64_t)r * kBlock;
      rid_[r] = index < n ? input[index] : 0u;
    }
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
      const uint64_t index = b * 1024 + tb + (uint64_t)r * kBlock;
      off_[r] = len_[r] = o_[r] = sv_[r] = 0;
      if (index < n) {
        off_[r] = indptr[rid_[r]];
        len_[r] = indptr[rid_[r] + 1] - off_[r];
        o_[r] = offset[index];
        sv_[r] = sm.value(rid_[r], index);
      }
    }
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
      const uint64_t index = b * 1024 + tb + (uint64_t)r * kBlock;
      if (index >= n) break;
      const uint32_t rid = rid_[r], off = off_[r], len = len_[r], o = o_[r], sv = sv_[r];
      if (len <= fanout) {
        for (uint32_t j = lig; j < len; j += G) {
          const uint32_t nbr = indices[off + j];
          out_src[o + j] = sv;
          out_dst[o + j] = nbr;
          if (di.w) di.enter(nbr, o + j); // direct dedup table of the batch: entered where it is produced
        }
        continue;
      }
      drew = true;
      uint32_t got = 0, tries = 0;
      const uint32_t len_magic = (uint32_t)(4294967296.0 / (double)len); // floor(2^32 / len), exact (len > fanout >= 1)
      while (got < fanout) {
        const Xorwow st0 = st;
        uint32_t kraw[R] = {}, uraw[R] = {};
#pragma unroll
        for (uint32_t i = 0; i < R; ++i) {
#pragma unroll
          for (uint32_t l = 0; l < G; ++l) { // try G i + l of the round: position draw, then acceptance draw
            const uint32_t a = st.next(), u = st.next();
            kraw[i] = (l == lig) ? a : kraw[i];
            uraw[i] = (l == lig) ? u : uraw[i];
          }
        }
        uint32_t cand[R];
        {
          uint32_t nb[R], al[R];
          float pr[R];
#pragma unroll
          for (uint32_t i = 0; i < R; ++i) { // one round trip: the alias is loaded unconditionally, with the rest
            // kraw % len: q' = mulhi(x, floor(2^32 / len)) is q or q - 1 (two fix-ups for safety)
            uint32_t k = kraw[i] - __umulhi(kraw[i], len_magic) * len;
            k = min(k, k - len);
            k = min(k, k - len);
            nb[i] = indices[off + k];
            al[i] = alias[off + k];
            pr[i] = prob[off + k];
          }
#pragma unroll
          for (uint32_t i = 0; i < R; ++i) cand[i] = (Xorwow::to_uniform(uraw[i]) > pr[i]) ? al[i] : nb[i]; // strict >, :101-103
        }
        // read-only probe: a candidate that is in the table now is in it whenever its turn comes
        bool need[R];
#pragma unroll
        for (uint32_t i = 0; i < R; ++i) {
          uint32_t pos = cand[i] % kDedupSlots, gap = 1;
          bool present = false;
          for (uint32_t probe = 0; probe < kDedupMaxProbes; ++probe) {
            if (tr[pos] != rid) break;
            if (tv[pos] == cand[i]) { present = true; break; }
            pos = (pos + gap) % kDedupSlots;
            ++gap;
          }
          need[i] = !present;
        }
        // the others, in try order, through insert_hash_table (:41-57) as the reference runs it
        const bool all_serial = tries + G * R > kDedupMaxTries; // the give-up rule looks at every try: no shortcut
        uint32_t used = G * R;
        bool done = false;
#pragma unroll
        for (uint32_t i = 0; i < R; ++i) {
          uint32_t m = done ? 0u : (uint32_t)(__ballot(need[i] || all_serial) >> grp_shift) & GMASK;
          while (m != 0) {
            const uint32_t t = (uint32_t)__ffs(m) - 1u;
            m &= m - 1u;
            const uint32_t c = __shfl(cand[i], (int)(grp_shift + t), 64);
            const bool give_up = tries + G * i + t + 1 > kDedupMaxTries;
            uint32_t pos = c % kDedupSlots, gap = 1;
            bool is_new = true; // a probe sequence that finds neither a free slot nor the value takes the candidate
            for (uint32_t probe = 0; probe < kDedupMaxProbes; ++probe) {
              if (tr[pos] != rid) {
                if (lig == 0) { tr[pos] = rid; tv[pos] = c; }
                break;
              }
              if (tv[pos] == c) { is_new = false; break; }
              pos = (pos + gap) % kDedupSlots;
              ++gap;
            }
            __builtin_amdgcn_wave_barrier();
            if (is_new || give_up) {
              if (lig == 0) {
                out_src[o + got] = sv;
                out_dst[o + got] = c;
                picked[g][got] = c;
              }
              ++got;
              if (got == fanout) {
                used = G * i + t + 1;
                done = true;
                m = 0;
              }
            }
          }
        }
        if (done) {
          if (used < G * R) {
            // Two draws per try actually made: the state after m = 2 * used draws.  The generator's v-array is a
            // sliding window over the sequence (old v0..v4, then the raw xorshift word of draw 0, 1, ...), so
            // v_j = window element m + j: five lane reads instead of replaying m steps.  Draw k of the round is
            // try k / 2 = G i + l: lane l holds it in kraw[i] (k even) or uraw[i] (k odd); its raw word is the
            // draw minus the Weyl counter after k + 1 steps.
            const uint32_t m = 2 * used;
            uint32_t v[5];
#pragma unroll
            for (uint32_t j = 0; j < 5; ++j) {
              const uint32_t e = m + j;                       // window element, >= 2
              const uint32_t k = e >= 5 ? e - 5 : 0;          // draw of the round (if e >= 5)
              const uint32_t ti = (k >> 1) / G, tl = (k >> 1) % G;
              uint32_t mine = 0;
#pragma unroll
              for (uint32_t i = 0; i < R; ++i) mine = (i == ti) ? ((k & 1u) ? uraw[i] : kraw[i]) : mine;
              const uint32_t x = __shfl(mine, (int)(grp_shift + tl), 64);
              const uint32_t from_draw = x - (st0.d + (k + 1) * 362437u);
              uint32_t from_old = st0.v4;
              from_old = (e == 3) ? st0.v3 : from_old;
              from_old = (e == 2) ? st0.v2 : from_old;
              v[j] = (e >= 5) ? from_draw : from_old;
            }
            st.v0 = v[0]; st.v1 = v[1]; st.v2 = v[2]; st.v3 = v[3]; st.v4 = v[4];
            st.d = st0.d + m * 362437u;
          }
        } else {
          tries += G * R;
        }
      }
      // the seed's picks into the batch's dedup table, by all 16 lanes at once (an atomic per accepted pick inside
      // the serial loop above would put its round trip on the stream's critical path)
      if (di.w) {
        __builtin_amdgcn_wave_barrier(); // lane 0's LDS writes above come first: LDS operations of a wave stay in order
        for (uint32_t j = lig; j < fanout; j += G) di.enter(picked[g][j], o + j);
      }
    }
    if (drew && lig == 0) st.store(states + 6 * sid);
    __builtin_amdgcn_wave_barrier(); // the tables are re-initialised by the next pass
  }
}

int sample_weighted_hash_dedup_impl(const uint32_t *indptr, const uint32_t *indices, const float *prob,
                                    const uint32_t *alias, const uint32_t *input, size_t n_max, Count n,
                                    uint32_t fanout, uint32_t *out_src, uint32_t *out_dst, uint64_t *num_out_dev,
                                    uint32_t *states, uint32_t *workspace, const uint32_t *seed_local, int src_local,
                                    hipStream_t s, ScanArea *shared_scan, const DedupInsert *insert) {
  uint32_t *offset = workspace;
  const ScanArea sa = shared_scan ? *shared_scan : ScanArea{offset + n_max, false};
  int rc = tile_scan(MinDegFanout{indptr, input, fanout}, StoreWord{offset}, n_max, n, sa, nullptr, nullptr,
                     num_out_dev, s);
  if (rc != GGMS_OK) return rc;
  const SrcMode sm{seed_local, src_local};
  const DedupInsert di = insert ? *insert : DedupInsert{};
  const size_t blocks = (n_max + 1023) / 1024; // reference blocks of 256 streams
  // G = lanes that share a stream: 16; tries per round = G R with R = 1.  "About 1.5 x fanout tries per round, so that
  // an ordinary seed is done in one round trip" (R = 4 at fanout 25) was the first choice and measured WORSE: 128
  // generator steps per lane and round and 120 VGPRs (the large products layer's 5312 waves did not fit the chip at
  // once) against 32 steps and 71 VGPRs -- step 0.62 -> 0.52 ms on products with two or three round trips per seed
  // (profiles/r03_ab_hash_dedup_tries_per_round.txt; R = 2: 0.55, R = 3: 0.59).  Four lanes per stream with 16 tries
  // each (a quarter of the generator instructions per stream) were built and measured slower too (a wave then waits
  // for the slowest of 16 streams instead of 4), as were 8 and 32 lanes per stream
  // (profiles/r03_ab_hash_dedup_lanes_per_stream.txt).  Only <16, 1> is built.
  hipLaunchKernelGGL((k_weighted_hash_dedup<16, 1>), dim3((unsigned)std::min<size_t>((256 / (64 / 16)) * blocks, 8192)),
                     dim3(kWave), 0, s, indptr, indices, prob, alias, input, n, fanout, offset, out_src, out_dst, states,
                     sm, di);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // namespace ggms

using namespace ggms;

extern "C" {

size_t ggms_sample_weighted_workspace_bytes(size_t num_input, size_t fanout) {
  return weighted_ws_words(num_input, fanout) * sizeof(uint32_t);
}

int ggms_sample_weighted_khop(const ggms_graph_t *graph, const float *prob_table, const ggms_id_t *alias_table,
                              const ggms_id_t *input, size_t num_input, size_t fanout, ggms_id_t *out_src,
                              ggms_id_t *out_dst, uint64_t *num_out_dev, void *states, size_t num_states,
                              void *workspace, size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev && fanout > 0);
  GGMS_CHECK_ARG(graph->num_part == 0); // "this algorithm not support DistGraph engine", dist_loops.cc:171-172
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(prob_table && alias_table && input && out_src && out_dst && states && workspace);
  GGMS_CHECK_ARG((uint64_t)num_input * fanout < (1ull << 32));
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_weighted_workspace_bytes(num_input, fanout));
  const uint64_t tasks = (uint64_t)num_input * fanout;
  const uint64_t threads = tasks < kWeightedMaxThreads ? tasks : kWeightedMaxThreads;
  const uint64_t span = (threads + 255) / 256 * 256;
  GGMS_CHECK_ARG((span < tasks ? span : tasks) <= num_states); // assert(thread_id < num_random_states), :52
  return sample_weighted_impl(graph->indptr, graph->indices, prob_table, alias_table, input, num_input,
                              count_of(num_input), (uint32_t)fanout, out_src, out_dst, num_out_dev, (uint32_t *)states,
                              (uint32_t *)workspace, nullptr, 0, s, nullptr, graph->num_node);
}

int ggms_sample_khop1(const ggms_graph_t *graph, const ggms_id_t *input, size_t num_input, size_t fanout,
                      ggms_id_t *out_src, ggms_id_t *out_dst, uint64_t *num_out_dev, void *states,
                      size_t num_states, void *workspace, size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev && fanout > 0);
  GGMS_CHECK_ARG(graph->num_part == 0); // "this algorithm not support DistGraph engine", dist_loops.cc:167-168
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(input && out_src && out_dst && states && workspace);
  GGMS_CHECK_ARG((uint64_t)num_input * fanout < (1ull << 32));
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_weighted_workspace_bytes(num_input, fanout));
  const uint64_t tasks = (uint64_t)num_input * fanout;
  const uint64_t threads = tasks < kWeightedMaxThreads ? tasks : kWeightedMaxThreads;
  const uint64_t span = (threads + 255) / 256 * 256;
  GGMS_CHECK_ARG((span < tasks ? span : tasks) <= num_states); // assert(thread_id < num_random_states), khop1.cu:51
  return sample_weighted_impl(graph->indptr, graph->indices, nullptr, nullptr, input, num_input, count_of(num_input),
                              (uint32_t)fanout, out_src, out_dst, num_out_dev, (uint32_t *)states,
                              (uint32_t *)workspace, nullptr, 0, s, nullptr, graph->num_node);
}

int ggms_sample_weighted_khop_prefix(const ggms_graph_t *graph, const float *prob_prefix_table, const ggms_id_t *input,
                                     size_t num_input, size_t fanout, ggms_id_t *out_src, ggms_id_t *out_dst,
                                     uint64_t *num_out_dev, void *states, size_t num_states, void *workspace,
                                     size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev && fanout > 0);
  GGMS_CHECK_ARG(graph->num_part == 0); // "this algorithm not support DistGraph engine", dist_loops.cc:209-210
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(prob_prefix_table && input && out_src && out_dst && states && workspace);
  GGMS_CHECK_ARG((uint64_t)num_input * fanout < (1ull << 32));
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_weighted_workspace_bytes(num_input, fanout));
  const uint64_t tasks = (uint64_t)num_input * fanout;
  const uint64_t threads = tasks < kWeightedMaxThreads ? tasks : kWeightedMaxThreads;
  const uint64_t span = (threads + 255) / 256 * 256;
  GGMS_CHECK_ARG((span < tasks ? span : tasks) <= num_states); // prefix.cu:50
  return sample_weighted_impl(graph->indptr, graph->indices, prob_prefix_table, nullptr, input, num_input,
                              count_of(num_input), (uint32_t)fanout, out_src, out_dst, num_out_dev, (uint32_t *)states,
                              (uint32_t *)workspace, nullptr, 0, s, nullptr, graph->num_node);
}

int ggms_sample_weighted_khop_hash_dedup(const ggms_graph_t *graph, const float *prob_table,
                                         const ggms_id_t *alias_table, const ggms_id_t *input, size_t num_input,
                                         size_t fanout, ggms_id_t *out_src, ggms_id_t *out_dst, uint64_t *num_out_dev,
                                         void *states, size_t num_states, void *workspace, size_t workspace_bytes,
                                         ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev && fanout > 0 && fanout < kDedupSlots);
  GGMS_CHECK_ARG(graph->num_part == 0); // dist_loops.cc:227-228
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(prob_table && alias_table && input && out_src && out_dst && states && workspace);
  GGMS_CHECK_ARG((uint64_t)num_input * fanout < (1ull << 32));
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_workspace_bytes(GGMS_WEIGHTED_KHOP_HASH_DEDUP, num_input, fanout));
  GGMS_CHECK_ARG((num_input + 1023) / 1024 * 256 <= num_states); // assert(i < num_random_states), hash_dedup.cu:70
  return sample_weighted_hash_dedup_impl(graph->indptr, graph->indices, prob_table, alias_table, input, num_input,
                                         count_of(num_input), (uint32_t)fanout, out_src, out_dst, num_out_dev,
                                         (uint32_t *)states, (uint32_t *)workspace, nullptr, 0, s, nullptr, nullptr);
}

} // extern "C"
