// micro_atomics.hip -- diagnostic: random-address atomic / load / store rates on a node-indexed table.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_atomics tools/micro_atomics.hip && /tmp/micro_atomics
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <typename W>
__global__ __launch_bounds__(256) void k_min(W *tab, const uint32_t *keys, uint32_t n, W hi) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicMin(&tab[keys[i]], (W)(hi | i));
}
template <typename W>
__global__ __launch_bounds__(256) void k_min_ret(W *tab, const uint32_t *keys, uint32_t n, W hi, uint32_t *out) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    out[i] = (uint32_t)atomicMin(&tab[keys[i]], (W)(hi | i));
}
template <typename W>
__global__ __launch_bounds__(256) void k_load(const W *tab, const uint32_t *keys, uint32_t n, uint32_t *out) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = (uint32_t)tab[keys[i]];
}
template <typename W>
__global__ __launch_bounds__(256) void k_store(W *tab, const uint32_t *keys, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) tab[keys[i]] = (W)i;
}
// workgroup-scope atomic (executes in the XCD's L2; NOT coherent across XCDs -- rate probe only)
template <typename W>
__global__ __launch_bounds__(256) void k_min_wg(W *tab, const uint32_t *keys, uint32_t n, W hi) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    __hip_atomic_fetch_min(&tab[keys[i]], (W)(hi | i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// XCD-local variant: block b only touches slice (b % 8) of the table (blocks are dealt to the 8 XCDs round-robin, so a
// slice of 1/8 of the table is what ONE XCD's 4-MiB L2 would see); keys are folded into the slice
template <typename W, bool WG>
__global__ __launch_bounds__(256) void k_min_sliced(W *tab, const uint32_t *keys, uint32_t n, W hi, uint32_t slice) {
  const uint32_t base = (blockIdx.x % 8) * slice;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    W *p = &tab[base + keys[i] % slice];
    if (WG) __hip_atomic_fetch_min(p, (W)(hi | i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicMin(p, (W)(hi | i));
  }
}
template <typename W>
__global__ __launch_bounds__(256) void k_load_sliced(const W *tab, const uint32_t *keys, uint32_t n, uint32_t *out,
                                                     uint32_t slice) {
  const uint32_t base = (blockIdx.x % 8) * slice;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = (uint32_t)tab[base + keys[i] % slice];
}

template <typename F>
static float time_us(F f, int reps = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / reps;
}

int main(int argc, char **argv) {
  const uint32_t N = argc > 1 ? atoi(argv[1]) : 2449029;  // table entries
  const uint32_t E = argc > 2 ? atoi(argv[2]) : 1860000;  // accesses
  std::vector<uint32_t> keys(E);
  uint64_t s = 88172645463325252ull;
  for (uint32_t i = 0; i < E; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    // mild skew: square of a uniform -> more mass on small ids, like degree-ordered hot nodes
    const double u = (double)(s >> 11) / 9007199254740992.0;
    keys[i] = (uint32_t)(u * u * N) % N;
  }
  uint32_t *d_keys, *d_out;
  void *d_tab;
  CK(hipMalloc(&d_keys, E * 4));
  CK(hipMalloc(&d_out, E * 4));
  CK(hipMalloc(&d_tab, (size_t)N * 8));
  CK(hipMemcpy(d_keys, keys.data(), E * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_tab, 0xff, (size_t)N * 8));
  for (int grid : {512, 2048, 8192}) {
    printf("grid %d blocks x 256, N=%u entries, E=%u accesses\n", grid, N, E);
    float t;
    t = time_us([&] { k_min<unsigned long long><<<grid, 256>>>((unsigned long long *)d_tab, d_keys, E, 0x1234ull << 32); });
    printf("  atomicMin u64 agent, no return : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_min<uint32_t><<<grid, 256>>>((uint32_t *)d_tab, d_keys, E, 0u); });
    printf("  atomicMin u32 agent, no return : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_min_ret<unsigned long long><<<grid, 256>>>((unsigned long long *)d_tab, d_keys, E, 0x1234ull << 32, d_out); });
    printf("  atomicMin u64 agent, returning : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_min_ret<uint32_t><<<grid, 256>>>((uint32_t *)d_tab, d_keys, E, 0u, d_out); });
    printf("  atomicMin u32 agent, returning : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_min_wg<unsigned long long><<<grid, 256>>>((unsigned long long *)d_tab, d_keys, E, 0x1234ull << 32); });
    printf("  atomicMin u64 workgroup scope  : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_min_wg<uint32_t><<<grid, 256>>>((uint32_t *)d_tab, d_keys, E, 0u); });
    printf("  atomicMin u32 workgroup scope  : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_load<unsigned long long><<<grid, 256>>>((const unsigned long long *)d_tab, d_keys, E, d_out); });
    printf("  load u64                       : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_load<uint32_t><<<grid, 256>>>((const uint32_t *)d_tab, d_keys, E, d_out); });
    printf("  load u32                       : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_store<unsigned long long><<<grid, 256>>>((unsigned long long *)d_tab, d_keys, E); });
    printf("  store u64                      : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_store<uint32_t><<<grid, 256>>>((uint32_t *)d_tab, d_keys, E); });
    printf("  store u32                      : %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    const uint32_t slice = N / 8;
    t = time_us([&] { k_min_sliced<unsigned long long, false><<<grid, 256>>>((unsigned long long *)d_tab, d_keys, E, 0x1234ull << 32, slice); });
    printf("  atomicMin u64 agent, XCD slices: %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_min_sliced<unsigned long long, true><<<grid, 256>>>((unsigned long long *)d_tab, d_keys, E, 0x1234ull << 32, slice); });
    printf("  atomicMin u64 wg,    XCD slices: %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_min_sliced<uint32_t, true><<<grid, 256>>>((uint32_t *)d_tab, d_keys, E, 0u, slice); });
    printf("  atomicMin u32 wg,    XCD slices: %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
    t = time_us([&] { k_load_sliced<unsigned long long><<<grid, 256>>>((const unsigned long long *)d_tab, d_keys, E, d_out, slice); });
    printf("  load u64,            XCD slices: %7.1f us  %6.1f G/s\n", t, E / t / 1e3);
  }
  return 0;
}
