// micro_atomics2.hip -- diagnostic: what limits random returning atomics -- the CUs' address path or the memory side?
//   rate vs number of CUs (hipExtStreamCreateWithCUMask), vs waves per CU, same-line pairs, beside random loads.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_atomics2 tools/micro_atomics2.hip && /tmp/micro_atomics2
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

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// OP 0: returning atomicMin u64; 1: 4-byte load; 2: both (load, then atomic on an independent key); 3: atomic, lanes paired
// on one 64-byte line (2 lanes -> adjacent words); 4: non-returning atomicMin
template <int OP>
__global__ __launch_bounds__(256) void k_rand(unsigned long long *tab, const uint32_t *ltab, uint32_t entries, uint32_t E,
                                              uint32_t salt, uint32_t *out) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < E; i += gridDim.x * 256) {
    uint32_t k = (uint32_t)(((uint64_t)mix(i * 2654435761u + salt) * entries) >> 32);
    if (OP == 3) k = ((uint32_t)(((uint64_t)mix((i >> 1) * 2654435761u + salt) * entries) >> 32) & ~7u) | (i & 1u);
    if (OP == 0 || OP == 3) acc += (uint32_t)atomicMin(&tab[k], ((unsigned long long)salt << 32) | i);
    if (OP == 4) atomicMin(&tab[k], ((unsigned long long)salt << 32) | i);
    if (OP == 1) acc += ltab[k];
    if (OP == 2) {
      const uint32_t k2 = (uint32_t)(((uint64_t)mix(i * 40503u + salt) * entries) >> 32);
      acc += ltab[k2];
      acc += (uint32_t)atomicMin(&tab[k], ((unsigned long long)salt << 32) | i);
    }
  }
  if (acc == 0x12345678u) *out = acc;
}

static float run(int op, hipStream_t s, int grid, unsigned long long *tab, const uint32_t *ltab, uint32_t entries, uint32_t E,
                 uint32_t *out) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int REP = 5;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) CK(hipEventRecord(a, s));
    for (int r = 0; r < REP; ++r) {
      const uint32_t salt = 0x7fffff00u - (uint32_t)(pass * REP + r) - (uint32_t)op * 64;
      switch (op) {
        case 0: k_rand<0><<<grid, 256, 0, s>>>(tab, ltab, entries, E, salt, out); break;
        case 1: k_rand<1><<<grid, 256, 0, s>>>(tab, ltab, entries, E, salt, out); break;
        case 2: k_rand<2><<<grid, 256, 0, s>>>(tab, ltab, entries, E, salt, out); break;
        case 3: k_rand<3><<<grid, 256, 0, s>>>(tab, ltab, entries, E, salt, out); break;
        default: k_rand<4><<<grid, 256, 0, s>>>(tab, ltab, entries, E, salt, out); break;
      }
    }
    if (pass) CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
  }
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / REP;
}

int main() {
  const uint32_t entries = 111059956u, E = 3400000u; // the papers100M dedup table, one batch's edges
  unsigned long long *tab;
  uint32_t *ltab, *out;
  CK(hipMalloc(&tab, (size_t)entries * 8));
  CK(hipMalloc(&ltab, (size_t)entries * 4));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(tab, 0xff, (size_t)entries * 8));
  CK(hipMemset(ltab, 1, (size_t)entries * 4));
  const char *names[] = {"atomicMin u64 returning", "load u32", "load + atomic", "atomic, lane pairs on one line", "atomicMin u64 no return"};
  hipStream_t s0;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  printf("-- all CUs, by grid (blocks of 256)\n");
  for (int grid : {256, 512, 1024, 2048, 4096, 8192})
    for (int op = 0; op < 5; ++op) {
      const float us = run(op, s0, grid, tab, ltab, entries, E, out);
      printf("grid %5d %-32s %8.1f us  %6.1f G/s\n", grid, names[op], us, E / us / 1e3);
    }
  // CU masks: `per` CUs of every XCD (32 CUs per XCD, 8 XCDs: mask bit = cu index as the runtime numbers them)
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("-- CU masks (device has %d CUs)\n", ncu);
  for (int frac : {8, 4, 2, 1}) { // 1/frac of the CUs, spread evenly
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    int used = 0;
    for (int c = 0; c < ncu; ++c)
      if (c % frac == 0) { mask[c / 32] |= 1u << (c % 32); ++used; }
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    for (int op : {0, 1, 2}) {
      const float us = run(op, s, 2048, tab, ltab, entries, E, out);
      printf("CUs %3d (every %d-th) %-32s %8.1f us  %6.1f G/s\n", used, frac, names[op], us, E / us / 1e3);
    }
    CK(hipStreamDestroy(s));
  }
  return 0;
}
