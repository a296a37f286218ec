// common.hip -- error plumbing, XORWOW state pool.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <random>

#include "tile_scan.h"

namespace ggms {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// one status word per device, allocated and zeroed on first use
uint32_t *device_status_word() {
  static std::mutex mu;
  static uint32_t *words[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!words[dev]) {
    uint32_t *p = nullptr;
    if (hipMalloc((void **)&p, 64) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 64) != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    words[dev] = p;
  }
  return words[dev];
}

// LDS one workgroup may use on the current device (gfx950: 160 KB), 0 if it cannot be read
size_t device_lds_bytes() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) return 0;
  return (size_t)v;
}

// Several kernels keep their working set in more than the default 64 KB of dynamic LDS (khop3 at large fan-outs,
// the random-walk top-K tile, the one-workgroup sort).  They are written for gfx950's 160 KB; on a device with
// less the request fails HERE with a message instead of as a bare launch error.
int raise_dynamic_lds(const void *func, size_t bytes, const char *who) {
  const size_t have = device_lds_bytes();
  if (have != 0 && bytes > have) {
    set_error("%s needs %zu bytes of LDS per workgroup, this device offers %zu: the library is built for gfx950 (160 KB)",
              who, bytes, have);
    return GGMS_ERR_INVALID;
  }
  GGMS_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return GGMS_OK;
}

bool host_readable_table(const void *table, const char *what) {
  static thread_local const void *last_ok[2] = {nullptr, nullptr}; // the hot loop hands over the same tables every batch
  if (table == last_ok[0] || table == last_ok[1]) return true;
  struct Remember {
    const void *t;
    bool ok = true;
    ~Remember() { if (ok) { last_ok[1] = last_ok[0]; last_ok[0] = t; } }
  } remember{table};
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, table) != hipSuccess) { // an ordinary host pointer the runtime has never seen
    (void)hipGetLastError();
    return true;
  }
  if (a.type == hipMemoryTypeDevice) {
    set_error("%s is a DEVICE pointer: since ABI 3 the shard pointer tables are HOST arrays (include/ggms.h)", what);
    remember.ok = false;
    return false;
  }
  return true;
}

std::atomic<long long> &debug_knob_word(int knob) {
  static std::atomic<long long> v[GGMS_DEBUG_NUM_KNOBS] = {{-1}, {-1}, {-1}};
  return v[knob];
}
long long debug_knob(int knob) { return debug_knob_word(knob).load(std::memory_order_relaxed); }

unsigned long long next_dedup_tag() {
  static std::atomic<unsigned long long> g{[] {
    std::random_device rd;
    return (((unsigned long long)rd() << 32) | rd()) | 1ull;
  }()};
  return g.fetch_add(1) + 1;
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

int ggms_abi_version(void) { return 3; }

void ggms_debug_set_knob(int knob, long long value) {
  if (knob >= 0 && knob < GGMS_DEBUG_NUM_KNOBS) debug_knob_word(knob).store(value < 0 ? -1 : value);
}

int ggms_device_status(uint32_t *status_host, int clear) {
  GGMS_CHECK_ARG(status_host);
  uint32_t *w = device_status_word();
  if (!w) {
    set_error("ggms_device_status: no device status word (no device?)");
    return GGMS_ERR_NO_DEVICE;
  }
  // kernels OR into the word from non-blocking streams, which a null-stream copy does not wait for: drain the device
  GGMS_HIP(hipDeviceSynchronize());
  GGMS_HIP(hipMemcpy(status_host, w, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (clear && *status_host) {
    GGMS_HIP(hipMemset(w, 0, sizeof(uint32_t)));
    GGMS_HIP(hipDeviceSynchronize());
  }
  return GGMS_OK;
}

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
      if (num_random_walk == 0) return 0; // the block-shape rule below never ends on 0 walks (callers refuse them)
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
