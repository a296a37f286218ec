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

constexpr uint32_t kMaxVisits = 128; // num_walk * walk_length supported per seed (LDS budget)

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

// One lane per seed: distinct visited nodes + counts in first-visit order, then K stable arg-max picks.
__global__ __launch_bounds__(kWave) void k_walk_topk(const uint32_t *__restrict__ tmp_src,
                                                     const uint32_t *__restrict__ tmp_dst, Count n_arg, uint64_t stride,
                                                     uint32_t per,
                                                     uint32_t K, uint32_t *__restrict__ pad_dst,
                                                     uint32_t *__restrict__ pad_cnt, uint32_t *__restrict__ num_top) {
  extern __shared__ uint32_t lds[]; // uniq[per][64], cnt[per][64]
  uint32_t *uniq = lds, *cnt = lds + per * kWave;
  const uint64_t n = n_arg.get();
  const uint32_t lane = threadIdx.x;
  for (uint64_t s = (uint64_t)blockIdx.x * kWave + lane; s < n; s += (uint64_t)gridDim.x * kWave) {
    uint32_t nu = 0;
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
          if (uniq[u * kWave + lane] == d) break;
        if (u == nu) {
          uniq[nu * kWave + lane] = d;
          cnt[nu * kWave + lane] = 1;
          ++nu;
        } else {
          cnt[u * kWave + lane] += 1;
        }
      }
    }
    const uint32_t take = nu < K ? nu : K;
    for (uint32_t k = 0; k < take; ++k) {
      uint32_t best = 0, best_c = 0;
      for (uint32_t u = 0; u < nu; ++u) {
        const uint32_t c = cnt[u * kWave + lane];
        if (c > best_c) { best_c = c; best = u; } // strict: ties keep the earlier visit
      }
      pad_dst[s * K + k] = uniq[best * kWave + lane];
      pad_cnt[s * K + k] = best_c;
      cnt[best * kWave + lane] = 0; // taken
    }
    num_top[s] = take;
  }
}

struct TopCount {
  const uint32_t *num_top;
  __device__ __forceinline__ uint32_t operator()(uint64_t s) const { return num_top[s]; }
};
struct TopEmit { // compact_output_revised, cuda_frequency_hashmap.cu:500-532
  const uint32_t *input, *pad_dst, *pad_cnt;
  uint32_t K;
  uint32_t *out_src, *out_dst, *out_data;
  const uint32_t *seed_local;
  int src_local;
  __device__ __forceinline__ void operator()(uint64_t s, uint32_t take, uint32_t at) const {
    const uint32_t sv = src_local ? (seed_local ? seed_local[s] : (uint32_t)s) : input[s];
    for (uint32_t k = 0; k < take; ++k) {
      out_src[at + k] = sv;
      out_dst[at + k] = pad_dst[s * K + k];
      out_data[at + k] = pad_cnt[s * K + k];
    }
  }
};

static void walk_block_shape(uint32_t num_walk, uint32_t &bx, uint32_t &by) {
  bx = kBlock; by = 1; // dim3 block(kCudaBlockSize, 1); while (x >= 2 * num_walk) { x /= 2; y *= 2; }  (:132-136)
  while (bx >= 2 * num_walk) { bx /= 2; by *= 2; }
}

size_t random_walk_ws_words(size_t num_input, size_t walk_length, size_t num_walk, size_t K) {
  return 2 * num_input * walk_length * num_walk + 2 * num_input * K + num_input + tile_scan_words(num_input) + 64;
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

int sample_random_walk_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t walk_length,
                            double restart_prob, uint32_t num_walk, uint32_t K, uint32_t *out_src, uint32_t *out_dst,
                            uint32_t *out_data, uint64_t *num_out_dev, uint32_t *states, uint32_t *workspace,
                            const uint32_t *seed_local, int src_local, hipStream_t s, ScanArea *shared_scan) {
  const uint32_t per = walk_length * num_walk;
  uint32_t *w = workspace;
  uint32_t *tmp_src = w; w += n_max * per;
  uint32_t *tmp_dst = w; w += n_max * per;
  uint32_t *pad_dst = w; w += n_max * K;
  uint32_t *pad_cnt = w; w += n_max * K;
  uint32_t *num_top = w; w += n_max;
  uint32_t *scan_scr = w;
  int rc = random_walk_raw_impl(g, input, n_max, n, walk_length, restart_prob, num_walk, tmp_src, tmp_dst, states, s);
  if (rc != GGMS_OK) return rc;
  hipLaunchKernelGGL(k_walk_topk, dim3(grid_for(n_max, kWave)), dim3(kWave), 2 * per * kWave * sizeof(uint32_t), s,
                     tmp_src, tmp_dst, n, (uint64_t)n_max, per, K, pad_dst, pad_cnt, num_top);
  GGMS_LAUNCH_CHECK();
  const ScanArea sa = shared_scan ? *shared_scan : ScanArea{scan_scr, false};
  return tile_scan(TopCount{num_top}, TopEmit{input, pad_dst, pad_cnt, K, out_src, out_dst, out_data, seed_local, src_local},
                   n_max, n, sa, nullptr, nullptr, num_out_dev, s);
}

} // namespace ggms

using namespace ggms;

extern "C" {

size_t ggms_sample_random_walk_workspace_bytes(size_t num_input, size_t walk_length, size_t num_walk, size_t K) {
  return random_walk_ws_words(num_input, walk_length, num_walk, K) * sizeof(uint32_t);
}

// PredictRandomWalkMaxThreads, cuda_random_states.cu:48-60
size_t ggms_random_walk_num_states(size_t num_input, size_t num_walk) {
  uint32_t bx, by;
  walk_block_shape((uint32_t)num_walk, bx, by);
  return (num_input + by - 1) / by * bx * by;
}

int ggms_sample_random_walk(const ggms_graph_t *graph, const ggms_id_t *input, size_t num_input, size_t walk_length,
                            double restart_prob, size_t num_walk, size_t K, ggms_id_t *out_src, ggms_id_t *out_dst,
                            ggms_id_t *out_data, uint64_t *num_out_dev, void *states, size_t num_states,
                            void *workspace, size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev && walk_length > 0 && num_walk > 0 && K > 0);
  GGMS_CHECK_ARG(walk_length * num_walk <= kMaxVisits && num_walk <= kBlock);
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(input && out_src && out_dst && out_data && states && workspace);
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_random_walk_workspace_bytes(num_input, walk_length, num_walk, K));
  GGMS_CHECK_ARG(ggms_random_walk_num_states(num_input, num_walk) <= num_states); // assert(thread_id < num_random_states)
  return sample_random_walk_impl(view_of(graph), input, num_input, count_of(num_input), (uint32_t)walk_length,
                                 restart_prob, (uint32_t)num_walk, (uint32_t)K, out_src, out_dst, out_data, num_out_dev,
                                 (uint32_t *)states, (uint32_t *)workspace, nullptr, 0, s);
}

} // extern "C"
