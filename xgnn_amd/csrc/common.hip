// common.hip -- error plumbing, tile-prefix kernel, XORWOW state pool.
#include <cstdarg>
#include <cstdio>

#include "tile_scan.h"

namespace ggms {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ __launch_bounds__(kBlock) void k_tile_prefix(const uint32_t *tile_sums, Count n_arg,
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

// cuda_random_states.cu:36-46
__global__ __launch_bounds__(kBlock) void k_init_states(uint32_t *states, uint64_t num, uint64_t seed) {
  for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < num; t += (uint64_t)gridDim.x * kBlock) {
    Xorwow st;
    st.init(seed + t);
    st.store(states + 6 * t);
  }
}

} // namespace ggms

using namespace ggms;

extern "C" {

int ggms_abi_version(void) { return 1; }

const char *ggms_last_error(void) { return g_err; }

size_t ggms_dtype_bytes(int dtype) {
  switch (dtype) {
    case GGMS_F32: return 4;
    case GGMS_F64: return 8;
    case GGMS_F16: return 2;
    case GGMS_U8: return 1;
    case GGMS_I32: return 4;
    case GGMS_I8: return 1;
    case GGMS_I64: return 8;
    default: return 0;
  }
}

int ggms_random_states_init(void *states, size_t num_states, uint64_t seed, ggms_stream_t stream) {
  GGMS_CHECK_ARG(states != nullptr || num_states == 0);
  if (num_states == 0) return GGMS_OK;
  hipLaunchKernelGGL(k_init_states, dim3(grid_for(num_states, kBlock)), dim3(kBlock), 0, to_stream(stream),
                     (uint32_t *)states, (uint64_t)num_states, seed);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

// PredictNumNodes, common.cc:488-497
static size_t predict_num_nodes(size_t batch, const size_t *fanout, size_t k) {
  size_t count = batch;
  for (int i = (int)k - 1; i >= 0; --i) count += count * fanout[i];
  return count;
}

// GPURandomStates ctor sizing, cuda_random_states.cu:48-97
size_t ggms_random_states_count(int sample_type, const size_t *fanout, size_t num_fanout, size_t batch_size,
                                size_t num_random_walk) {
  const size_t kMaxThreads = 512 * 1024; // constant.h:71-73
  switch (sample_type) {
    case GGMS_KHOP0:
    case GGMS_KHOP2:
    case GGMS_KHOP3:
    case GGMS_WEIGHTED_KHOP_HASH_DEDUP:
      return predict_num_nodes(batch_size, fanout, num_fanout - 1);
    case GGMS_KHOP1:
    case GGMS_WEIGHTED_KHOP:
    case GGMS_WEIGHTED_KHOP_PREFIX: {
      size_t n = predict_num_nodes(batch_size, fanout, num_fanout);
      return n < kMaxThreads ? n : kMaxThreads;
    }
    case GGMS_RANDOM_WALK: {
      size_t nodes = predict_num_nodes(batch_size, fanout, num_fanout - 1);
      size_t bx = 256, by = 1;
      while (bx >= 2 * num_random_walk) { bx /= 2; by *= 2; }
      size_t grid = (nodes + by - 1) / by;
      return grid * bx * by;
    }
    default:
      return 0;
  }
}

} // extern "C"
