// micro_mall.hip -- diagnostic: what a random-access table costs BESIDE a streaming copy, by table size.
//
// The pipelined step runs the HBM-bound row gather next to the sampler's random accesses (dedup-table atomics,
// table stores, neighbour reads).  Measured in bench.py: the gather alone 523 us, the sampler alone 484 us, both
// together 795 us -- far more than the ~12 % of extra HBM bytes the sampler adds.  This probe separates the
// candidates: does a table that fits the 256-MiB Infinity Cache keep its accesses off the DRAM while a stream of
// non-temporal loads/stores runs past it?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_mall tools/micro_mall.hip && /tmp/micro_mall
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// streaming copy, 16 B per lane, non-temporal both ways (what the row gather does)
template <bool NT>
__global__ __launch_bounds__(256) void k_stream(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    else dst[i] = src[i];
  }
}

// E random accesses into a table of `entries` 8-byte words; salt changes the key set per launch
template <int OP> // 0 = atomicMin, 1 = load, 2 = store
__global__ __launch_bounds__(256) void k_random(unsigned long long *tab, uint32_t entries, uint32_t E, uint32_t salt,
                                                uint32_t *sink) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < E; i += gridDim.x * 256) {
    const uint32_t k = (uint32_t)(((uint64_t)mix(i * 2654435761u + salt) * entries) >> 32);
    if (OP == 0) atomicMin(&tab[k], ((unsigned long long)salt << 32) | i);
    else if (OP == 1) acc += (uint32_t)tab[k];
    else tab[k] = i;
  }
  if (OP == 1 && acc == 0x12345678u) *sink = acc;
}

static float elapsed_us(hipEvent_t a, hipEvent_t b) {
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f;
}

int main() {
  const size_t stream_bytes = (size_t)3 << 30; // per launch: read 3 GiB / 2, write the same (like one batch's gather)
  const size_t n16 = stream_bytes / 2 / 16;
  const uint32_t E = 3400000; // random accesses per launch (one batch's edges)
  u32x4 *src, *dst;
  uint32_t *sink;
  CK(hipMalloc(&src, n16 * 16));
  CK(hipMalloc(&dst, n16 * 16));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(src, 1, n16 * 16));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t a1, b1, a2, b2;
  CK(hipEventCreate(&a1)); CK(hipEventCreate(&b1)); CK(hipEventCreate(&a2)); CK(hipEventCreate(&b2));
  const int REP = 6;
  auto run_stream = [&](hipStream_t s, bool nt) {
    for (int r = 0; r < REP; ++r) {
      if (nt) k_stream<true><<<2048, 256, 0, s>>>(src, dst, n16);
      else k_stream<false><<<2048, 256, 0, s>>>(src, dst, n16);
    }
  };
  // stream alone
  for (int nt = 1; nt >= 0; --nt) {
    run_stream(s1, nt);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a1, s1));
    run_stream(s1, nt);
    CK(hipEventRecord(b1, s1));
    CK(hipDeviceSynchronize());
    const float us = elapsed_us(a1, b1) / REP;
    printf("stream alone (%s): %.1f us per launch, %.2f TB/s\n", nt ? "nt" : "plain", us, stream_bytes / us / 1e6);
  }
  const size_t table_mb[] = {16, 64, 128, 192, 512, 900};
  const char *opname[] = {"atomicMin", "load", "store"};
  for (size_t mb : table_mb) {
    const uint32_t entries = (uint32_t)((mb << 20) / 8);
    unsigned long long *tab;
    CK(hipMalloc(&tab, (size_t)entries * 8));
    CK(hipMemset(tab, 0xff, (size_t)entries * 8));
    for (int op = 0; op < 3; ++op) {
      auto run_random = [&](hipStream_t s, int launches, uint32_t salt0) {
        for (int r = 0; r < launches; ++r) {
          const uint32_t salt = salt0 + r;
          if (op == 0) k_random<0><<<2048, 256, 0, s>>>(tab, entries, E, 0x7fffffffu - salt, sink);
          else if (op == 1) k_random<1><<<2048, 256, 0, s>>>(tab, entries, E, salt, sink);
          else k_random<2><<<2048, 256, 0, s>>>(tab, entries, E, salt, sink);
        }
      };
      // random alone
      run_random(s2, REP, 1);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a2, s2));
      run_random(s2, REP, 100);
      CK(hipEventRecord(b2, s2));
      CK(hipDeviceSynchronize());
      const float alone = elapsed_us(a2, b2) / REP;
      // both: REP stream launches on s1; random launches on s2 for about as long (4 x REP, the alone time is shorter)
      const int rl = 4 * REP;
      CK(hipEventRecord(a1, s1));
      CK(hipEventRecord(a2, s2));
      run_stream(s1, true);
      run_random(s2, rl, 1000);
      CK(hipEventRecord(b1, s1));
      CK(hipEventRecord(b2, s2));
      CK(hipDeviceSynchronize());
      const float st = elapsed_us(a1, b1) / REP, rt = elapsed_us(a2, b2) / rl;
      printf("table %4zu MB %-9s: alone %6.1f us (%5.1f G/s) | beside the nt stream: %6.1f us per launch, stream %6.1f us "
             "per launch (%.2f TB/s)\n", mb, opname[op], alone, E / alone / 1e3, rt, st, stream_bytes / st / 1e6);
    }
    CK(hipFree(tab));
  }
  return 0;
}
