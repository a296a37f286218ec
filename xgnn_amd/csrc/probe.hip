// probe.hip -- rate probe of the memory side for the bench report (bench.py `roofline_sampler`).
//
// The sampler's per-edge work is one random 4-byte neighbour load and one returning 64-bit atomicMin on a word of the
// node-indexed dedup table.  Both are bounded at the memory side, not in the CUs: tools/micro_atomics2.hip measures
// the same rates on 32 CUs as on 256 and with 1 to 32 waves per CU, and two lanes that hit one 64-byte line cost one
// request (profiles/r03_micro_atomics2.txt).  This entry point lets the bench measure those ceilings in its own
// process, on a table of the size the run uses, so that "the chain is N x its request floor" can be checked from
// the JSON line alone.
#include "ggms_internal.h"

namespace ggms {

__device__ __forceinline__ uint32_t probe_mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_fabric_probe(unsigned long long *table, size_t table_words, uint32_t n,
                                                         uint32_t salt, uint32_t *sink) {
  const uint32_t *as32 = reinterpret_cast<const uint32_t *>(table);
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const uint32_t k = (uint32_t)(((uint64_t)probe_mix(i * 2654435761u + salt) * table_words) >> 32);
    if (KIND == GGMS_PROBE_LOAD || KIND == GGMS_PROBE_LOAD_ATOMIC) {
      const uint32_t k2 = (uint32_t)(((uint64_t)probe_mix(i * 40503u + ~salt) * table_words) >> 32);
      acc += as32[2 * (size_t)k2 + 1]; // high half: the atomics below only ever lower a word
    }
    if (KIND == GGMS_PROBE_ATOMIC || KIND == GGMS_PROBE_LOAD_ATOMIC)
      acc += (uint32_t)atomicMin(&table[k], ((unsigned long long)salt << 32) | i);
  }
  if (acc == 0x9e3779b9u) *sink = acc; // keeps the loads alive
}

} // namespace ggms

using namespace ggms;

extern "C" {

int ggms_fabric_probe(int kind, void *table, size_t table_words, size_t num_requests, uint32_t salt, void *sink,
                      ggms_stream_t stream) {
  GGMS_CHECK_ARG(table && sink && table_words > 0 && table_words < (1ull << 32));
  GGMS_CHECK_ARG(num_requests > 0 && num_requests < (1ull << 32));
  GGMS_CHECK_ARG(kind == GGMS_PROBE_ATOMIC || kind == GGMS_PROBE_LOAD || kind == GGMS_PROBE_LOAD_ATOMIC);
  const dim3 grid(grid_for(num_requests, kBlock)), block(kBlock);
  unsigned long long *t = (unsigned long long *)table;
  hipStream_t s = to_stream(stream);
  if (kind == GGMS_PROBE_ATOMIC)
    hipLaunchKernelGGL(k_fabric_probe<GGMS_PROBE_ATOMIC>, grid, block, 0, s, t, table_words, (uint32_t)num_requests, salt, (uint32_t *)sink);
  else if (kind == GGMS_PROBE_LOAD)
    hipLaunchKernelGGL(k_fabric_probe<GGMS_PROBE_LOAD>, grid, block, 0, s, t, table_words, (uint32_t)num_requests, salt, (uint32_t *)sink);
  else
    hipLaunchKernelGGL(k_fabric_probe<GGMS_PROBE_LOAD_ATOMIC>, grid, block, 0, s, t, table_words, (uint32_t)num_requests, salt, (uint32_t *)sink);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // extern "C"
