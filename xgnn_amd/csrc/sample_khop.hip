// sample_khop.hip -- uniform fan-out neighbour samplers (khop3, khop0).
//
// Reference: GPUSampleKHop3 (cuda/cuda_sampling_khop3.cu:76-146, host :234-318) and
// GPUSampleKHop0 NEW_ALGO (cuda/cuda_sampling_khop0.cu:102-153, host :243-335).
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
//     insert per draw, khop3.cu:125-131).  Here ONE lane owns one RNG stream
//     ("group"), i.e. 64 independent streams per wave64, each with a private
//     open-addressing set in LDS laid out lane-interleaved (bank = lane % 32,
//     conflict-free for ds_read_b32/ds_write_b32);
//   * khop0: one lane per logical reservoir lane (32 per seed, as the RNG stream
//     assignment demands); the racy atomicExch (khop0.cu:144-148) becomes an LDS
//     atomicMax on the candidate position, i.e. highest-j-wins, deterministic.
#include "tile_scan.h"

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

// What goes into out_src: the seed's global id (leaf API, like the reference) or
// its local id (fused batch path: the COO `col` is written by the sampler itself).
struct SrcMode {
  const uint32_t *seed_local; // local id per seed position; NULL = the position itself
  int local;
  __device__ __forceinline__ uint32_t value(uint32_t rid, uint64_t index) const {
    if (!local) return rid;
    return seed_local ? seed_local[index] : (uint32_t)index;
  }
};

// ---- khop3 -------------------------------------------------------------------
// Group (b, y) of the reference grid == stream id i = 8 b + y; it serves seeds
// 128 b + y + 8 k, k = 0..15, in that order (khop3.cu:86-89,106).
//
// Two launches:
//   k_khop3_positions  one lane per stream: the serial part (RNG + rejection),
//                      no neighbour loads; writes the chosen POSITION r_j into
//                      out_dst[o + j].  The 16 seeds' (id, degree, offset) chains
//                      are fetched up front into LDS so their latency is paid once.
//   k_gather_neighbours  flat over edges: out_dst[e] = neighbours(seed)[out_dst[e]].
// Keeping the random 4-byte neighbour reads out of the serial loop is what
// matters: in one fused loop each lane waited ~1 us per neighbour.
template <int SET_BITS>
__global__ __launch_bounds__(kWave) void k_khop3_positions(GraphView g, const uint32_t *__restrict__ input,
                                                           Count n_arg, uint32_t fanout,
                                                           const uint32_t *__restrict__ offset,
                                                           uint32_t *__restrict__ out_src,
                                                           uint32_t *__restrict__ out_dst,
                                                           uint32_t *__restrict__ states, SrcMode sm) {
  constexpr uint32_t SLOTS = 1u << SET_BITS;
  constexpr uint32_t SMASK = SLOTS - 1;
  constexpr uint32_t HASH_EMPTY = 0xffffffffu;
  __shared__ uint32_t set_tab[SLOTS * kWave]; // [slot][lane]
  __shared__ uint8_t set_used[128 * kWave];   // slot of the j-th insert of the current seed, [j][lane]
  __shared__ uint32_t seed_len[16 * kWave];   // [k][lane]
  __shared__ uint32_t seed_off[16 * kWave];
  __shared__ uint32_t seed_src[16 * kWave];
  const uint64_t n = n_arg.get();
  const uint32_t lane = threadIdx.x;
  const uint64_t num_groups = ((n + 127) / 128) * 8;

  for (uint32_t s = 0; s < SLOTS; ++s) set_tab[s * kWave + lane] = HASH_EMPTY;

  for (uint64_t grp = (uint64_t)blockIdx.x * kWave + lane; grp < num_groups; grp += (uint64_t)gridDim.x * kWave) {
    const uint64_t b = grp >> 3, y = grp & 7;
    Xorwow st;
    st.load(states + 6 * grp);
    uint32_t rid[16];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
      const uint64_t index = 128 * b + y + 8 * k;
      rid[k] = index < n ? input[index] : kEmptyKey;
    }
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
      const uint64_t index = 128 * b + y + 8 * k;
      uint32_t len = 0, off = 0, sv = 0;
      if (index < n) {
        g.neighbours(rid[k], len);
        off = offset[index];
        sv = sm.value(rid[k], index);
      }
      seed_len[k * kWave + lane] = len;
      seed_off[k * kWave + lane] = off;
      seed_src[k * kWave + lane] = sv;
    }
    for (uint32_t k = 0; k < 16; ++k) {
      const uint64_t index = 128 * b + y + 8 * k;
      if (index >= n) break;
      const uint32_t len = seed_len[k * kWave + lane];
      const uint32_t o = seed_off[k * kWave + lane];
      const uint32_t sv = seed_src[k * kWave + lane];
      if (len <= fanout) {
        for (uint32_t j = 0; j < len; ++j) {
          out_src[o + j] = sv;
          out_dst[o + j] = j;
        }
      } else {
        uint32_t count = 0;
        while (count < fanout) {
          const uint32_t r = st.next() % len;
          uint32_t pos = r & SMASK, delta = 1;
          for (;;) {
            const uint32_t cur = set_tab[pos * kWave + lane];
            if (cur == HASH_EMPTY) {
              set_tab[pos * kWave + lane] = r;
              set_used[count * kWave + lane] = (uint8_t)pos;
              out_src[o + count] = sv;
              out_dst[o + count] = r; // insertion order == output order (items[] of khop3.cu:64)
              ++count;
              break;
            }
            if (cur == r) break;
            pos = (pos + delta) & SMASK;
            ++delta;
          }
        }
        for (uint32_t j = 0; j < fanout; ++j) // leave the set empty for the next seed
          set_tab[(uint32_t)set_used[j * kWave + lane] * kWave + lane] = HASH_EMPTY;
      }
    }
    st.store(states + 6 * grp);
  }
}

// out_dst[e] holds a position inside the seed's neighbour list; replace it by the neighbour id.
// The seed comes from out_src[e]: its global id (leaf API) or, in local mode, n2o[out_src[e]].
__global__ __launch_bounds__(kBlock) void k_gather_neighbours(GraphView g, const uint32_t *__restrict__ out_src,
                                                              uint32_t *__restrict__ out_dst,
                                                              const uint64_t *__restrict__ num_out,
                                                              const uint32_t *__restrict__ local_to_global) {
  const uint64_t n = *num_out;
  for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (uint64_t)gridDim.x * kBlock) {
    const uint32_t sv = out_src[e];
    const uint32_t rid = local_to_global ? local_to_global[sv] : sv;
    uint32_t len;
    const uint32_t *edges = g.neighbours(rid, len);
    out_dst[e] = edges[out_dst[e]];
  }
}

// ---- khop0 (reservoir) -------------------------------------------------------
// Block = 128 threads = 4 logical warps of 32 lanes; thread t: x = t % 32, w = t / 32.
// Stream seed = (b*128 + x*4 + w) + num_input (khop0.cu:114-117).
__global__ __launch_bounds__(128) void k_sample_khop0(GraphView g, const uint32_t *__restrict__ input,
                                                      Count n_arg, uint32_t fanout,
                                                      const uint32_t *__restrict__ offset,
                                                      uint32_t *__restrict__ out_src,
                                                      uint32_t *__restrict__ out_dst, SrcMode sm) {
  extern __shared__ uint32_t slot_j[]; // [4][fanout]: winning position per reservoir slot
  const uint64_t n = n_arg.get();
  const uint32_t x = threadIdx.x & 31, w = threadIdx.x >> 5;
  uint32_t *my_slots = slot_j + w * fanout;
  const uint64_t num_blocks = (n + 63) / 64;
  for (uint64_t b = blockIdx.x; b < num_blocks; b += gridDim.x) {
    Xorwow st;
    st.init((uint64_t)(b * 128 + x * 4 + w) + n);
    const uint64_t last = (64 * (b + 1) < n) ? 64 * (b + 1) : n;
    for (uint64_t index = 64 * b + w; index < last; index += 4) {
      const uint32_t rid = input[index];
      uint32_t len;
      const uint32_t *edges = g.neighbours(rid, len);
      const uint32_t o = offset[index];
      const uint32_t sv = sm.value(rid, index);
      if (len <= fanout) {
        for (uint32_t j = x; j < len; j += 32) {
          out_src[o + j] = sv;
          out_dst[o + j] = edges[j];
        }
      } else {
        uint32_t j = x;
        for (; j < fanout; j += 32) my_slots[j] = j; // slot j starts as position j
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (; j < len; j += 32) {
          const uint32_t kk = st.next() % (j + 1);
          if (kk < fanout) atomicMax(&my_slots[kk], j); // highest j wins
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t s = x; s < fanout; s += 32) {
          out_src[o + s] = sv;
          out_dst[o + s] = edges[my_slots[s]];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

size_t sample_ws_words(size_t num_input) { return num_input + tile_scan_words(num_input) + 16; }

// offsets by exclusive scan of min(deg, fanout), then the sampler proper
int sample_khop3_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t fanout, uint32_t *out_src,
                      uint32_t *out_dst, uint64_t *num_out_dev, uint32_t *states, uint32_t *workspace,
                      const uint32_t *seed_local, int src_local, const uint32_t *local_to_global, hipStream_t s) {
  uint32_t *offset = workspace;
  uint32_t *scratch = offset + n_max;
  int rc = tile_scan(SeedCount{g, input, fanout}, StoreOffset{offset}, n_max, n, scratch, nullptr, nullptr,
                     num_out_dev, s);
  if (rc != GGMS_OK) return rc;
  const size_t num_groups = (n_max + 127) / 128 * 8;
  const int grid = grid_for(num_groups, kWave);
  const SrcMode sm{seed_local, src_local};
  if (fanout < 32) {
    hipLaunchKernelGGL((k_khop3_positions<6>), dim3(grid), dim3(kWave), 0, s, g, input, n, fanout, offset, out_src,
                       out_dst, states, sm);
  } else {
    // 128 slots, the reference's HASHTABLE_SIZE (khop3.cu:43): load factor < 0.5 up to fanout 63
    hipLaunchKernelGGL((k_khop3_positions<7>), dim3(grid), dim3(kWave), 0, s, g, input, n, fanout, offset, out_src,
                       out_dst, states, sm);
  }
  GGMS_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gather_neighbours, dim3(grid_for(n_max * fanout, kBlock)), dim3(kBlock), 0, s, g, out_src,
                     out_dst, num_out_dev, src_local ? local_to_global : nullptr);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

int sample_khop0_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t fanout, uint32_t *out_src,
                      uint32_t *out_dst, uint64_t *num_out_dev, uint32_t *workspace, const uint32_t *seed_local,
                      int src_local, hipStream_t s) {
  uint32_t *offset = workspace;
  uint32_t *scratch = offset + n_max;
  int rc = tile_scan(SeedCount{g, input, fanout}, StoreOffset{offset}, n_max, n, scratch, nullptr, nullptr,
                     num_out_dev, s);
  if (rc != GGMS_OK) return rc;
  const size_t num_blocks = (n_max + 63) / 64;
  hipLaunchKernelGGL(k_sample_khop0, dim3(grid_for(num_blocks, 1)), dim3(128), 4 * fanout * sizeof(uint32_t), s, g,
                     input, n, fanout, offset, out_src, out_dst, SrcMode{seed_local, src_local});
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // namespace ggms

using namespace ggms;

extern "C" {

size_t ggms_sample_workspace_bytes(int sample_type, size_t num_input, size_t fanout) {
  (void)sample_type;
  (void)fanout;
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
  return sample_khop3_impl(view_of(graph), input, num_input, count_of(num_input), (uint32_t)fanout, out_src, out_dst,
                           num_out_dev, (uint32_t *)states, (uint32_t *)workspace, nullptr, 0, nullptr, s);
}

int ggms_sample_khop0(const ggms_graph_t *graph, const ggms_id_t *input, size_t num_input, size_t fanout,
                      ggms_id_t *out_src, ggms_id_t *out_dst, uint64_t *num_out_dev, void *workspace,
                      size_t workspace_bytes, ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && num_out_dev);
  GGMS_CHECK_ARG(fanout > 0 && fanout <= 8192);
  hipStream_t s = to_stream(stream);
  if (num_input == 0) {
    GGMS_HIP(hipMemsetAsync(num_out_dev, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  GGMS_CHECK_ARG(input && out_src && out_dst && workspace);
  GGMS_CHECK_ARG(workspace_bytes >= ggms_sample_workspace_bytes(GGMS_KHOP0, num_input, fanout));
  GGMS_CHECK_ARG((uint64_t)num_input * fanout < (1ull << 32));
  return sample_khop0_impl(view_of(graph), input, num_input, count_of(num_input), (uint32_t)fanout, out_src, out_dst,
                           num_out_dev, (uint32_t *)workspace, nullptr, 0, s);
}

} // extern "C"
