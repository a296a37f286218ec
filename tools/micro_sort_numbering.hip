// micro_sort_numbering.hip -- probe, not product: could a batch's ordered dedup be done WITHOUT one HBM atomic per
// sampled edge?  Candidate: stable radix sort of the batch's (node id, index) pairs + run heads + a flag scatter and an
// ordered scan in the batch's own index space (3.4 M items: <= 27 MB, Infinity-Cache resident) + one scatter of the
// local ids.  Against it: what the product does today -- one returning 64-bit atomicMin per edge on the 0.9-GB
// direct table (the figure ggms_fabric_probe reports in every bench line) -- alone and beside a 3-GB copy stream (the
// gather that runs next to the sampler in the pipeline).
//
//   sort path   k_sort_hist / scan / k_sort_scatter x 4 passes (8-bit digits, 27-bit node ids; xgnn_amd/csrc/radix_sort.h,
//               the tree's own sort), k_heads (run head = first occurrence: flag[index] = 1, head position per item),
//               tile_scan over the flags in index order (local id of every owner), k_spread (row[index] = local id of
//               the item's run head)
//   atomics     k_atomics: one returning atomicMin per pair on a table of one 64-bit word per node id
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o /tmp/micro_sort_numbering tools/micro_sort_numbering.hip \
//       -Lxgnn_amd/lib -lggms_hip -Wl,-rpath,$PWD/xgnn_amd/lib && /tmp/micro_sort_numbering
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../xgnn_amd/csrc/radix_sort.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

using namespace ggms;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// a batch's neighbour ids: E draws from a pool of U distinct nodes (U < E: duplicates as in a sampled frontier)
__global__ void k_make(uint32_t *keys, uint32_t *vals, uint32_t E, uint32_t U, uint32_t N, uint32_t salt) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < E; i += gridDim.x * 256) {
    const uint32_t u = (uint32_t)(((uint64_t)mix(i * 2654435761u + salt) * U) >> 32);
    keys[i] = (uint32_t)(((uint64_t)mix(u * 40503u + 7u) * N) >> 32);
    vals[i] = i;
  }
}

// sorted order: item p is a run head if its key differs from its left neighbour's.  Heads own their key (stable sort:
// the head carries the smallest index of the run).  flag[index] = 1 for heads; head_pos[p] = position of p's run head.
__global__ void k_heads(const uint32_t *keys, const uint32_t *vals, uint32_t E, uint32_t *flag, uint32_t *head_of) {
  for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < E; p += gridDim.x * 256) {
    const uint32_t k = keys[p];
    uint32_t h = p;
    while (h > 0 && keys[h - 1] == k) --h; // runs are short (a few instances per node)
    head_of[p] = vals[h];
    flag[vals[p]] = (h == p) ? 1u : 0u;
  }
}
struct FlagValue {
  const uint32_t *flag;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return flag[i]; }
};
struct StoreLocal {
  uint32_t *local;
  __device__ __forceinline__ void operator()(uint64_t i, uint32_t, uint32_t excl) const { local[i] = excl; }
};
// every instance gets the local id of its run head
__global__ void k_spread(const uint32_t *vals, const uint32_t *head_of, const uint32_t *local, uint32_t E, uint32_t *row) {
  for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < E; p += gridDim.x * 256) row[vals[p]] = local[head_of[p]];
}

__global__ void k_atomics(unsigned long long *tab, const uint32_t *keys, uint32_t E, uint32_t salt, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < E; i += gridDim.x * 256)
    acc += (uint32_t)atomicMin(&tab[keys[i]], ((unsigned long long)salt << 32) | i);
  if (acc == 0x12345678u) *sink = acc;
}

__global__ void k_stream(const uint4 *src, uint4 *dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main() {
  const uint32_t N = 111059956u, E = 3400000u, U = 3000000u;
  uint32_t *k0, *v0, *k1, *v1, *flag, *head_of, *local, *row, *scratch, *sink;
  unsigned long long *tab;
  CK(hipMalloc(&k0, E * 4)); CK(hipMalloc(&v0, E * 4)); CK(hipMalloc(&k1, E * 4)); CK(hipMalloc(&v1, E * 4));
  CK(hipMalloc(&flag, E * 4)); CK(hipMalloc(&head_of, E * 4)); CK(hipMalloc(&local, E * 4)); CK(hipMalloc(&row, E * 4));
  CK(hipMalloc(&sink, 4));
  const size_t sw = sort_scratch_words(E) + tile_scan_words(E) + 64;
  CK(hipMalloc(&scratch, sw * 4));
  CK(hipMalloc(&tab, (size_t)N * 8));
  CK(hipMemset(tab, 0xff, (size_t)N * 8));
  const size_t stream_bytes = 1500ull << 20; // read 1.5 GB + write 1.5 GB per launch: a papers100M batch's gather
  uint4 *sa, *sb;
  CK(hipMalloc(&sa, stream_bytes)); CK(hipMalloc(&sb, stream_bytes));
  CK(hipMemset(sa, 1, stream_bytes));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));

  auto sort_path = [&](uint32_t salt) {
    hipLaunchKernelGGL(k_make, dim3(1024), dim3(256), 0, s, k0, v0, E, U, N, salt);
    return 0;
  };
  (void)sort_path;
  uint32_t *scan_words = scratch + sort_scratch_words(E);
  auto numbering = [&]() -> int {
    bool second = false;
    int rc = radix_sort_pairs(k0, v0, k1, v1, E, count_of(E), scratch, s, N, &second);
    if (rc != GGMS_OK) return rc;
    const uint32_t *ks = second ? k1 : k0, *vs = second ? v1 : v0;
    hipLaunchKernelGGL(k_heads, dim3(2048), dim3(256), 0, s, ks, vs, E, flag, head_of);
    rc = tile_scan(FlagValue{flag}, StoreLocal{local}, E, count_of(E), ScanArea{scan_words, false}, nullptr, nullptr, nullptr, s);
    if (rc != GGMS_OK) return rc;
    hipLaunchKernelGGL(k_spread, dim3(2048), dim3(256), 0, s, vs, head_of, local, E, row);
    return GGMS_OK;
  };

  for (int beside = 0; beside < 2; ++beside) {
    float t_sort = 0, t_atom = 0;
    const int REP = 5;
    for (int which = 0; which < 2; ++which) {
      for (int pass = 0; pass < 2; ++pass) { // pass 0 warms up
        float total = 0;
        for (int r = 0; r < REP; ++r) {
          const uint32_t salt = 0x7ffff000u - (uint32_t)(which * 64 + pass * REP + r);
          hipLaunchKernelGGL(k_make, dim3(1024), dim3(256), 0, s, k0, v0, E, U, N, salt); // fresh keys (the sort consumes k0)
          CK(hipStreamSynchronize(s));
          if (beside) hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, s2, sa, sb, stream_bytes / 16);
          CK(hipEventRecord(a, s));
          if (which == 0) {
            if (numbering() != GGMS_OK) { fprintf(stderr, "numbering failed: %s\n", ggms_last_error()); return 1; }
          } else {
            hipLaunchKernelGGL(k_atomics, dim3(2048), dim3(256), 0, s, tab, k0, E, salt, sink);
          }
          CK(hipEventRecord(b, s));
          CK(hipDeviceSynchronize());
          float ms;
          CK(hipEventElapsedTime(&ms, a, b));
          total += ms;
        }
        if (pass) (which == 0 ? t_sort : t_atom) = total / REP;
      }
    }
    printf("%s: sort-based numbering %.1f us (4 x {histogram, scan, scatter} + heads + ordered scan + spread), "
           "atomics %.1f us (%.1f G/s)  ->  atomics / sort = %.2f\n",
           beside ? "beside a 3-GB copy stream" : "alone                    ", t_sort * 1e3, t_atom * 1e3, E / (t_atom * 1e-3) / 1e9,
           t_atom / t_sort);
  }
  // sanity: the numbering is a numbering (row ids dense, first occurrences ascending)
  std::vector<uint32_t> h_row(E), h_key(E);
  hipLaunchKernelGGL(k_make, dim3(1024), dim3(256), 0, s, k0, v0, E, U, N, 0x1234u);
  CK(hipMemcpyAsync(h_key.data(), k0, E * 4, hipMemcpyDeviceToHost, s));
  if (numbering() != GGMS_OK) return 1;
  CK(hipMemcpyAsync(h_row.data(), row, E * 4, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  uint32_t next = 0, bad = 0;
  std::vector<uint32_t> seen; // local id -> key
  for (uint32_t i = 0; i < E && bad < 5; ++i) {
    if (h_row[i] == next) { seen.push_back(h_key[i]); ++next; }
    else if (h_row[i] > next || seen[h_row[i]] != h_key[i]) ++bad;
  }
  printf("check: %u unique of %u, %s\n", next, E, bad ? "NUMBERING WRONG" : "first-occurrence numbering ok");
  return bad ? 1 : 0;
}
