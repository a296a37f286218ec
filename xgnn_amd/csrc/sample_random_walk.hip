// sample_random_walk.hip -- PinSAGE neighbourhood: random walks with restart + per-seed top-K by visit count.
//
// Reference: GPUSampleRandomWalk (cuda/cuda_sampling_random_walk.cu:43-165) followed by
// FrequencyHashmap::GetTopK (cuda/cuda_frequency_hashmap.cu:643-841): seven phases -- per-seed open-addressed
// edge tables in HBM, count kernel, two device scans, unique-edge list, a 64-bit descending radix sort of
// ((num_seed - seed_idx) << 32 | count), per-seed prefix kernels, compaction, two table-reset kernels -- with
// eight host syncs.
//
// The result is, per seed position: the distinct visited nodes, ordered by count descending with ties in
// first-visit order (the stable sort over a list emitted in ascending winner index, :377-425,733-746), cut
// at K; seeds in input order; (src = seed id, dst = visited, data = count).  A seed has at most
// num_walk * walk_length visits (12 for the PinSAGE defaults), so the whole top-K of a seed fits in one
// lane's registers + a few LDS words: ONE kernel walks and ranks, one ordered scan compacts.
// Kept bit-for-bit: which XORWOW stream serves which (seed, walk) -- state index
// bx*by*block + by*walk + node_in_block with (bx, by) from the reference's block-shape rule (:132-136) --
// and three draws per step (curand % deg, then curand_uniform_double = two draws).
#include "ggms_internal.h"
#include "tile_scan.h"

namespace ggms {

constexpr uint32_t kLdsVisits = 128; // num_walk * walk_length up to which a seed's visits are ranked in LDS

__global__ __launch_bounds__(kBlock) void k_random_walk(GraphView g, const uint32_t *__restrict__ input, Count n_arg,
                                                        uint32_t walk_length, double restart_prob, uint32_t num_walk,
                                                        uint32_t bx, uint32_t by, uint64_t stride,
                                                        uint32_t *__restrict__ tmp_src,
                                                        uint32_t *__restrict__ tmp_dst, uint32_t *__restrict__ states) {
  const uint64_t n = n_arg.get();
  // thread t of the block: walk lane tx = t / by, node lane ty = t % by  (state = bx*by*block + by*tx + ty, :50-52)
  const uint32_t tx = threadIdx.x / by, ty = threadIdx.x % by;
  const uint64_t num_blocks = (n + by - 1) / by;
  for (uint64_t blk = blockIdx.x; blk < num_blocks; blk += gridDim.x) {
    const uint64_t node_idx = blk * by + ty;
    const uint64_t sid = (uint64_t)bx * by * blk + (uint64_t)by * tx + ty;
    Xorwow st;
    st.load(states + 6 * sid);
    if (node_idx < n) {
      const uint32_t start = input[node_idx];
      for (uint32_t walk = tx; walk < num_walk; walk += bx) {
        uint32_t node = start;
        for (uint32_t step = 0; step < walk_length; ++step) {
          // visit e = step * num_walk + walk of the seed (the reference's [seed][step][walk] order, :84-86), stored
          // visit-major so that the lanes of a wave -- consecutive seeds -- write and later read consecutive words
          const uint64_t pos = ((uint64_t)step * num_walk + walk) * stride + node_idx;
          if (node == kEmptyKey) {
            tmp_src[pos] = kEmptyKey;
          } else {
            uint32_t len;
            const uint32_t *edges = g.neighbours(node, len);
            if (len == 0) {
              tmp_src[pos] = kEmptyKey;
              node = kEmptyKey;
            } else {
              const uint32_t k = st.next() % len;
              tmp_src[pos] = start;
              node = edges[k];
              tmp_dst[pos] = node;
              if (st.uniform_double() < restart_prob) node = kEmptyKey; // terminate, :98-100
            }
          }
        }
      }
    }
    st.store(states + 6 * sid);
  }
}

// The rest of the layer in ONE launch: a workgroup takes a tile of T consecutive seeds from a ticket, one lane per seed
//   1. collects the seed's distinct visited nodes + counts in first-visit order (its LDS column);
//   2. scans take = min(distinct, K) over the tile and publishes the tile aggregate (decoupled look-back,
//      tile_scan.h) -- compact_output_revised's offsets (cuda_frequency_hashmap.cu:500-532);
//   3. makes its K stable arg-max picks (strict >, so ties keep the earlier visit: the stable descending sort of
//      :733-746) and parks them at their place in the tile's compact slice, in LDS;
//   4. wave 0 looks back for the tile's base offset (the predecessors published long ago: the wait hides behind 3);
//   5. all lanes sweep the tile's slice edge by edge: (src, dst, count) written coalesced and -- INSERT, direct
//      dedup table -- the visited node entered into the table (DedupInsert::enter), one atomic per lane per round
//      instead of K dependent ones.
//
// SPILL (more than kLdsVisits visits per seed -- far beyond PinSAGE's 12; the reference has no such bound, its
// edge tables live in HBM): the lane's column is its own visit slice of tmp_dst / tmp_src, compacted in place (the
// u-th distinct node lands at or before the visit it was read from), and every lane writes its own picks.
// what tile t's descriptor holds -- sum over its T seeds of min(distinct visited nodes, K) -- computed by one wave from
// the walk's visit lists (scan_lookback's Help).  Not for SPILL launches: those compact the lists in place.
template <uint32_t T>
struct WalkTileHelp {
  static constexpr bool kCan = true;
  const uint32_t *tmp_src, *tmp_dst;
  uint64_t n, stride;
  uint32_t per, K;
  __device__ __forceinline__ uint32_t operator()(uint64_t t) const {
    uint32_t acc = 0;
    for (uint32_t j = 0; j < T / kWave; ++j) {
      const uint64_t s = t * T + j * kWave + (threadIdx.x & 63u);
      if (s >= n) continue;
      uint32_t nu = 0;
      for (uint32_t e = 0; e < per; ++e) {
        if (tmp_src[(uint64_t)e * stride + s] == kEmptyKey) continue;
        const uint32_t d = tmp_dst[(uint64_t)e * stride + s];
        bool seen = false;
        for (uint32_t f = 0; f < e; ++f)
          seen |= tmp_src[(uint64_t)f * stride + s] != kEmptyKey && tmp_dst[(uint64_t)f * stride + s] == d;
        nu += seen ? 0u : 1u;
      }
      acc += nu < K ? nu : K;
    }
    return wave_reduce_sum(acc);
  }
};

template <uint32_t T, bool INSERT, bool SPILL>
__global__ __launch_bounds__(T) void k_walk_topk_emit(uint32_t *tmp_src, uint32_t *tmp_dst, Count n_arg,
                                                      uint64_t stride, uint32_t per, uint32_t K, uint32_t Kc,
                                                      const uint32_t *__restrict__ input, SrcMode sm,
                                                      uint32_t *__restrict__ out_src, uint32_t *__restrict__ out_dst,
                                                      uint32_t *__restrict__ out_data, FusedScan fs, DedupInsert di) {
  constexpr uint32_t FLAG_A = 1, FLAG_P = 2, W = T / kWave;
  extern __shared__ uint32_t lds[]; // uniq[per][T], cnt[per][T], stage_dst[Kc * T], stage_cnt[Kc * T]
  uint32_t *const uniq = SPILL ? tmp_dst : lds, *const cnt = SPILL ? tmp_src : lds + per * T;
  uint32_t *const stage_dst = lds + 2 * per * T, *const stage_cnt = stage_dst + Kc * T; // !SPILL only
  __shared__ uint32_t s_wsum[W], s_prefix;
  __shared__ uint64_t s_tile;
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + T - 1) / T;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  // A launch with a workgroup per tile (every layer but the largest ones) needs no ticket: tile = workgroup id, one
  // memory round trip less on the tile's latency chain.  Nothing is assumed about when a predecessor's workgroup runs:
  // a look-back that has waited long enough computes the missing aggregates itself (WalkTileHelp).  SPILL launches
  // rewrite their input, so they keep the strict ticket instead (a taken tile's predecessors are running).
  const bool one_tile_each = !SPILL && gridDim.x >= num_tiles; // uniform
  for (uint32_t turn = 0;; ++turn) {
    if (one_tile_each) {
      if (turn != 0) break;
    } else {
      if (tid == 0) s_tile = take_ticket(fs.tick, SPILL);
      __syncthreads();
    }
    const uint64_t b = one_tile_each ? (uint64_t)blockIdx.x : s_tile;
    if (b >= num_tiles) break;
    const uint64_t s = b * T + tid;
    const auto col = [&](uint32_t u) -> uint64_t { return SPILL ? (uint64_t)u * stride + s : (uint64_t)u * T + tid; };
    // ---- 1
    uint32_t nu = 0;
    if (s < n) {
      for (uint32_t e0 = 0; e0 < per; e0 += 8) { // 16 independent loads in flight, then the serial bookkeeping
        uint32_t sv[8], dv[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
          const uint64_t idx = (uint64_t)(e0 + k) * stride + s;
          sv[k] = (e0 + k < per) ? tmp_src[idx] : kEmptyKey;
          dv[k] = (e0 + k < per) ? tmp_dst[idx] : 0u; // tmp_dst of an empty visit is never written: value unused
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
          if (sv[k] == kEmptyKey) continue;
          const uint32_t d = dv[k];
          uint32_t u = 0;
          for (; u < nu; ++u)
            if (uniq[col(u)] == d) break;
          if (u == nu) {
            uniq[col(nu)] = d;
            cnt[col(nu)] = 1;
            ++nu;
          } else {
            cnt[col(u)] += 1;
          }
        }
      }
    }
    // ---- 2
    const uint32_t take = nu < K ? nu : K;
    const uint32_t incl = wave_inclusive_scan(take);
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t i = 0; i < W; ++i) {
      const uint32_t x = s_wsum[i];
      if (i < wv) before += x;
      total += x;
    }
    if (tid == 0)
      __hip_atomic_store(&fs.desc[b], scan_desc(fs.epoch, b == 0 ? FLAG_P : FLAG_A, total), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    // ---- 3
    const uint32_t at = before + incl - take;
    if (!SPILL)
      for (uint32_t k = 0; k < take; ++k) {
        uint32_t best = 0, best_c = 0;
        for (uint32_t u = 0; u < nu; ++u) {
          const uint32_t c = cnt[col(u)];
          if (c > best_c) { best_c = c; best = u; }
        }
        stage_dst[at + k] = uniq[col(best)];
        stage_cnt[at + k] = (tid << 8) | best_c; // count <= per <= kLdsVisits
        cnt[col(best)] = 0; // taken
      }
    // ---- 4
    if (tid < kWave) {
      uint32_t prefix = 0;
      if (b != 0) {
        if (SPILL) prefix = scan_lookback(fs.desc, b, fs.epoch, fs.err);
        else prefix = scan_lookback(fs.desc, b, fs.epoch, fs.err, fs.patience, WalkTileHelp<T>{tmp_src, tmp_dst, n, stride, per, K});
        if (lane == 0)
          __hip_atomic_store(&fs.desc[b], scan_desc(fs.epoch, FLAG_P, prefix + total), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        s_prefix = prefix;
        if (b + 1 == num_tiles) *fs.num_out = (uint64_t)prefix + total;
      }
    }
    __syncthreads();
    // ---- 5
    const uint32_t prefix = s_prefix;
    if (SPILL) {
      const uint32_t srcv = (s < n) ? (sm.local ? sm.value(0u, s) : input[s]) : 0u;
      for (uint32_t k = 0; k < take; ++k) {
        uint32_t best = 0, best_c = 0;
        for (uint32_t u = 0; u < nu; ++u) {
          const uint32_t c = cnt[col(u)];
          if (c > best_c) { best_c = c; best = u; }
        }
        const uint32_t e = prefix + at + k, d = uniq[col(best)];
        cnt[col(best)] = 0; // taken
        out_src[e] = srcv;
        out_dst[e] = d;
        out_data[e] = best_c;
        if (INSERT) di.enter(d, e);
      }
    } else {
      for (uint32_t o = tid; o < total; o += T) { // (four slots per lane and round measured no faster)
        const uint32_t d = stage_dst[o], pc = stage_cnt[o];
        const uint64_t index = b * T + (pc >> 8);
        const uint32_t e = prefix + o;
        out_src[e] = sm.local ? sm.value(0u, index) : input[index];
        out_dst[e] = d;
        out_data[e] = pc & 0xffu;
        if (INSERT) di.enter(d, e);
      }
    }
    __syncthreads(); // LDS is rewritten by the next tile
  }
  if (tid == 0 && num_tiles == 0 && blockIdx.x == 0) *fs.num_out = 0;
}

static void walk_block_shape(uint32_t num_walk, uint32_t &bx, uint32_t &by) {
  bx = kBlock; by = 1; // dim3 block(kCudaBlockSize, 1); while (x >= 2 * num_walk) { x /= 2; y *= 2; }  (:132-136)
  while (bx > 1 && bx >= 2 * num_walk) { bx /= 2; by *= 2; } // bx > 1: num_walk = 0 must not spin (callers refuse it)
}

// tile of the top-K kernel: the largest whose LDS (columns + staging) stays within 64 KB, down to one wave
// (kLdsVisits visits, K = visits: 128 KB); beyond that the spilling variant, which uses no LDS
static uint32_t walk_tile(uint32_t per, uint32_t Kc) {
  if (per + Kc <= 32) return 256;
  if (per + Kc <= 64) return 128;
  return 64;
}
size_t walk_scan_tiles(size_t num_input) { return (num_input + 63) / 64 + 2; }

size_t random_walk_ws_words(size_t num_input, size_t walk_length, size_t num_walk, size_t K) {
  (void)K;
  return 2 * num_input * walk_length * num_walk + 8 + 2 * walk_scan_tiles(num_input) + 64 + kTicketWords + 16;
}

// tmp_src / tmp_dst: [walk_length * num_walk][n_max], visit-major
int random_walk_raw_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t walk_length,
                         double restart_prob, uint32_t num_walk, uint32_t *tmp_src, uint32_t *tmp_dst,
                         uint32_t *states, hipStream_t s) {
  uint32_t bx, by;
  walk_block_shape(num_walk, bx, by);
  const size_t num_blocks = (n_max + by - 1) / by;
  hipLaunchKernelGGL(k_random_walk, dim3(grid_for(num_blocks, 1)), dim3(bx * by), 0, s, g, input, n, walk_length,
                     restart_prob, num_walk, bx, by, (uint64_t)n_max, tmp_src, tmp_dst, states);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

template <uint32_t T, bool INSERT, bool SPILL>
static int launch_topk_emit(int grid, size_t lds, hipStream_t s, uint32_t *tmp_src, uint32_t *tmp_dst,
                             Count n, uint64_t stride, uint32_t per, uint32_t K, uint32_t Kc, const uint32_t *input,
                             SrcMode sm, uint32_t *out_src, uint32_t *out_dst, uint32_t *out_data, FusedScan fs,
                             DedupInsert di) {
  if (lds > 48 * 1024) {
    static const int raised = raise_dynamic_lds(reinterpret_cast<const void *>(&k_walk_topk_emit<T, INSERT, SPILL>),
                                                160 * 1024 - 4096, "k_walk_topk_emit");
    if (raised != GGMS_OK) return raised;
  }
  hipLaunchKernelGGL((k_walk_topk_emit<T, INSERT, SPILL>), dim3(grid), dim3(T), lds, s, tmp_src, tmp_dst, n, stride, per, K,
                     Kc, input, sm, out_src, out_dst, out_data, fs, di);
  return GGMS_OK;
}

// shared_scan: the batch's scan area (cleared by the batch prologue); else the workspace holds a private one that
// is cleared here.  insert (direct dedup table): the visited nodes are entered into the table on the way out.
int sample_random_walk_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t walk_length,
                            double restart_prob, uint32_t num_walk, uint32_t K, uint32_t *out_src, uint32_t *out_dst,
                            uint32_t *out_data, uint64_t *num_out_dev, uint32_t *states, uint32_t *workspace,
                            const uint32_t *seed_local, int src_local, hipStream_t s, ScanArea *shared_scan,
                            const DedupInsert *insert) {
  const uint32_t per = walk_length * num_walk;
  uint32_t *w = workspace;
  uint32_t *tmp_src = w; w += n_max * per;
  uint32_t *tmp_dst = w; w += n_max * per;
  int rc = random_walk_raw_impl(g, input, n_max, n, walk_length, restart_prob, num_walk, tmp_src, tmp_dst, states, s);
  if (rc != GGMS_OK) return rc;
  const uint32_t Kc = K < per ? K : per;
  const uint32_t T = per > kLdsVisits ? 64u : walk_tile(per, Kc);
  const size_t tiles = (n_max + T - 1) / T;
  uint32_t *ctl = scan_align(shared_scan ? shared_scan->words : w);
  // tickets: the batch's next set (zeroed by the batch prologue); a private area keeps its set behind the descriptors
  uint32_t *tick = shared_scan ? take_ticket_set(shared_scan) : ctl + 8 + 2 * (tiles + 1) + 2;
  if (!tick) {
    set_error("sample_random_walk: the shared scan area has no ticket set left");
    return GGMS_ERR_INVALID;
  }
  if (!shared_scan)
    GGMS_HIP(hipMemsetAsync(ctl, 0, (8 + 2 * (tiles + 1) + 2 + kTicketWords) * sizeof(uint32_t), s));
  else if (!shared_scan->cleared)
    GGMS_HIP(hipMemsetAsync(ctl, 0, (8 + 2 * (tiles + 1)) * sizeof(uint32_t), s));
  const FusedScan fs{tick, reinterpret_cast<unsigned long long *>(ctl + 8), next_scan_epoch(), num_out_dev,
                     shared_scan ? shared_scan->status_word() : device_status_word(), scan_patience()};
  const SrcMode sm{seed_local, src_local};
  const bool spill = per > kLdsVisits;
  const size_t lds = spill ? 0 : 2 * (size_t)(per + Kc) * T * sizeof(uint32_t);
  const int grid = grid_for(tiles, 1);
  const DedupInsert none{};
  int rc_l = GGMS_OK;
#define GGMS_TOPK(TT, SP)                                                                                          \
  do {                                                                                                             \
    if (insert)                                                                                                    \
      rc_l = launch_topk_emit<TT, true, SP>(grid, lds, s, tmp_src, tmp_dst, n, (uint64_t)n_max, per, K, Kc, input, sm,    \
                                     out_src, out_dst, out_data, fs, *insert);                                     \
    else                                                                                                           \
      rc_l = launch_topk_emit<TT, false, SP>(grid, lds, s, tmp_src, tmp_dst, n, (uint64_t)n_max, per, K, Kc, input, sm,   \
                                      out_src, out_dst, out_data, fs, none);                                       \
  } while (0)
  if (spill) GGMS_TOPK(64, true);
  else if (T == 256) GGMS_TOPK(256, false);
  else if (T == 128) GGMS_TOPK(128, false);
  else GGMS_TOPK(64, false);
#undef GGMS_TOPK
  if (rc_l != GGMS_OK) return rc_l;
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // namespace ggms

using namespace ggms;

extern "C" {

size_t ggms_sample_random_walk_workspace_bytes(size_t num_input, size_t walk_length, size_t num_walk, size_t K) {
  return random_walk_ws_words(num_input, walk_length, num_walk, K) * sizeof(uint32_t);
}

// PredictRandomWalkMaxThreads, cuda_random_states.cu:48-60
size_t ggms_random_walk_num_states(size_t num_input, size_t num_walk) {
  if (num_walk == 0) return 0;
  uint32_t bx, by;
  walk_block_shape((uint32_t)num_walk, bx, by);
  return (num_input + by - 1) / by * bx * by;
}

int ggms_sample_random_walk(const ggms_graph_t *graph, const ggms_id_t *input, size_t num_input, size_t walk_length,
                            double restart_prob, size_t num_walk, size_t K, ggms_id_t *out_src, ggms_id_t *out_dst,
                            ggms_id_t *out_data, uint64_t *num_out_dev, void *states, size_t num_states,
                            void *workspace, size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev && walk_length > 0 && num_walk > 0 && K > 0);
  GGMS_CHECK_ARG(walk_length * num_walk < (1ull << 31));
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(input && out_src && out_dst && out_data && states && workspace);
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_random_walk_workspace_bytes(num_input, walk_length, num_walk, K));
  GGMS_CHECK_ARG(ggms_random_walk_num_states(num_input, num_walk) <= num_states); // assert(thread_id < num_random_states)
  GraphView gv;
  if (!view_of(graph, gv)) return GGMS_ERR_INVALID;
  return sample_random_walk_impl(gv, input, num_input, count_of(num_input), (uint32_t)walk_length,
                                 restart_prob, (uint32_t)num_walk, (uint32_t)K, out_src, out_dst, out_data, num_out_dev,
                                 (uint32_t *)states, (uint32_t *)workspace, nullptr, 0, s, nullptr, nullptr);
}

} // extern "C"
