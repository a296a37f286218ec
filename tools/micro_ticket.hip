// micro_ticket.hip -- diagnostic: what a global ticket costs.  G workgroups each take ONE returning atomicAdd on a
// shared counter (thread 0), optionally spread over S counters 64 bytes apart; against an empty kernel of the same grid
// (workgroup dispatch alone) and against per-workgroup atomics on private addresses.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_ticket tools/micro_ticket.hip && /tmp/micro_ticket
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

__global__ void k_empty(uint32_t *out) {
  if (out == nullptr && threadIdx.x == 9999) out[0] = 1;
}
// S counters, 16 words apart; block b uses counter b % S; `loops` tickets per block, one after the other
__global__ void k_ticket(uint32_t *ctr, uint32_t S, uint32_t loops, uint32_t *out) {
  __shared__ uint32_t s;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < loops; ++i) {
    if (threadIdx.x == 0) s = atomicAdd(&ctr[16 * (blockIdx.x % S)], 1u);
    __syncthreads();
    acc += s;
    __syncthreads();
  }
  if (acc == 0xffffffffu) out[0] = acc;
}
// fire-and-forget on one address (what a per-wave counter update is)
__global__ void k_noret(unsigned long long *ctr, uint32_t S, uint32_t loops) {
  for (uint32_t i = 0; i < loops; ++i)
    if ((threadIdx.x & 63u) == 0) atomicAdd(&ctr[8 * ((blockIdx.x * 4 + (threadIdx.x >> 6)) % S)], 1ull);
}

template <typename F>
static float time_us(F f, int reps = 10) {
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

int main() {
  uint32_t *ctr, *out;
  CK(hipMalloc(&ctr, 1 << 20));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(ctr, 0, 1 << 20));
  for (int block : {128, 256}) {
    for (int grid : {64, 256, 1024, 2048, 4096, 8192}) {
      const float e = time_us([&] { k_empty<<<grid, block>>>(out); });
      printf("block %3d grid %5d empty kernel                    %8.2f us\n", block, grid, e);
      for (uint32_t S : {1u, 8u, 32u, 128u}) {
        const float t = time_us([&] { k_ticket<<<grid, block>>>(ctr, S, 1, out); });
        printf("block %3d grid %5d 1 ticket per WG, %3u counters     %8.2f us  (%.1f ns per ticket over the empty kernel)\n", block, grid, S, t,
               (t - e) * 1e3 / grid);
      }
    }
  }
  // the fused sampler's pattern: 2048 workgroups, 3 tickets each, one after the other
  for (uint32_t S : {1u, 8u, 32u}) {
    const float t = time_us([&] { k_ticket<<<2048, 128>>>(ctr, S, 3, out); });
    printf("2048 WGs x 3 tickets in turn, %3u counters: %8.2f us\n", S, t);
  }
  // per-wave fire-and-forget counter updates: 46 K waves, 3 each
  for (uint32_t S : {1u, 8u, 64u}) {
    const float t = time_us([&] { k_noret<<<2048, 256>>>((unsigned long long *)ctr, S, 17); });
    printf("2048 x 4 waves x 17 no-return atomics (139 K), %3u counters: %8.2f us  (%.1f ns each)\n", S, t, t * 1e3 / (2048 * 4 * 17));
  }
  return 0;
}
