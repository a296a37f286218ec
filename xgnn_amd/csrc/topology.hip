// topology.hip -- link / topology probe of one node.
//
// Reference: PartitionSolver::DetectTopo / DetectTopo_child / LoadTopoFromFile (cuda/dist_graph.cu:684-726,
// 779-884, 886-938): a forked child that sees every GPU asks cudaDeviceCanAccessPeer for every pair, times one
// 128-MiB peer copy per reachable pair and writes a text file ("GPU Count", "P2P Matrix", "Bandwidth Matrix") that the
// placement solver reads back.
//
// Here: the same probe for one process that sees every GPU (ggms_detect_topology: the engine's forked probe child,
// engine.cc:DetectTopo), the same file format (ggms_topology_write_host / _read_host: a file written by either
// side loads on the other), and -- because the product runs one process PER GPU and reads its peers through hipIpc
// mappings -- the two measurements a rank can take on a mapping it already holds: a timed copy out of it
// (ggms_link_probe_copy) and the product's own gather kernel reading random rows of it (ggms_link_probe_gather: what
// the `peer` / `hybrid` feature stores do to xGMI every batch).
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>

#include "ggms_internal.h"

namespace ggms {

// index[i] = a pseudo-random row of [0, total): the probe's stand-in for a batch's slot list
__global__ __launch_bounds__(kBlock) void k_probe_index(uint32_t *__restrict__ index, uint64_t n, uint32_t total,
                                                        uint32_t seed) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    uint32_t x = (uint32_t)i * 2654435761u + seed; // one round of a 32-bit mix (murmur3 finaliser)
    x ^= x >> 16; x *= 0x85ebca6bu;
    x ^= x >> 13; x *= 0xc2b2ae35u;
    x ^= x >> 16;
    index[i] = (uint32_t)(((uint64_t)x * total) >> 32);
  }
}

// a plain streaming copy, 16 B per lane and 8 chunks in flight per lane (every load instruction of a wave covers one
// contiguous KiB): the kind of kernel behind the guide's device-to-device copy ceiling
typedef uint32_t u32x4_probe __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void k_probe_copy16(u32x4_probe *__restrict__ dst, const u32x4_probe *__restrict__ src,
                                                         uint64_t n) {
  constexpr int U = 8;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4_probe v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
  }
  for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

static int timed(hipStream_t s, int reps, const std::function<int()> &enqueue, double *seconds) {
  hipEvent_t e0, e1;
  GGMS_HIP(hipEventCreate(&e0));
  GGMS_HIP(hipEventCreate(&e1));
  int rc = enqueue(); // warm: first touch of a mapping, page tables, clocks
  if (rc == GGMS_OK) rc = hipEventRecord(e0, s) == hipSuccess ? GGMS_OK : GGMS_ERR_HIP;
  for (int r = 0; r < reps && rc == GGMS_OK; ++r) rc = enqueue();
  if (rc == GGMS_OK && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) {
    set_error("link probe: %s", hipGetErrorString(hipGetLastError()));
    rc = GGMS_ERR_HIP;
  }
  float ms = 0;
  if (rc == GGMS_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = GGMS_ERR_HIP;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *seconds = ms * 1e-3 / (reps > 0 ? reps : 1);
  return rc;
}

} // namespace ggms

using namespace ggms;

extern "C" {

int ggms_device_count(int *count) {
  GGMS_CHECK_ARG(count);
  GGMS_HIP(hipGetDeviceCount(count));
  return GGMS_OK;
}

int ggms_peer_access(int device, int peer, int *can_access) {
  GGMS_CHECK_ARG(can_access);
  if (device == peer) { // cudaDeviceCanAccessPeer refuses the diagonal; the reference's matrix holds 1 there (:808-811)
    *can_access = 1;
    return GGMS_OK;
  }
  GGMS_HIP(hipDeviceCanAccessPeer(can_access, device, peer));
  return GGMS_OK;
}

// DetectTopo_child, dist_graph.cu:779-848 -- for a process that sees all the GPUs (never a rank of the bench, never an
// engine worker: this call creates a context on EVERY device).
int ggms_detect_topology(ggms_topology_t *topo, size_t probe_bytes, int reps) {
  GGMS_CHECK_ARG(topo);
  std::memset(topo, 0, sizeof(*topo));
  int n = 0;
  GGMS_HIP(hipGetDeviceCount(&n));
  if (n > GGMS_TOPO_MAX_DEVICE) {
    set_error("ggms_detect_topology: %d devices, at most %d", n, GGMS_TOPO_MAX_DEVICE);
    return GGMS_ERR_INVALID;
  }
  topo->num_device = n;
  const size_t nbytes = probe_bytes ? probe_bytes : ((size_t)1 << 27); // 128 MiB, :783
  if (reps < 1) reps = 1;
  void *buf[GGMS_TOPO_MAX_DEVICE] = {nullptr}, *src[GGMS_TOPO_MAX_DEVICE] = {nullptr};
  hipStream_t st[GGMS_TOPO_MAX_DEVICE] = {nullptr};
  int rc = GGMS_OK;
  auto fail = [&](hipError_t e, const char *what, int d, int p) {
    set_error("ggms_detect_topology: %s (device %d, peer %d): %s", what, d, p, hipGetErrorString(e));
    rc = GGMS_ERR_HIP;
  };
  for (int d = 0; d < n && rc == GGMS_OK; ++d) {
    hipError_t e = hipSetDevice(d);
    if (e == hipSuccess) e = hipMalloc(&buf[d], nbytes);
    if (e == hipSuccess) e = hipMalloc(&src[d], nbytes);
    if (e == hipSuccess) e = hipMemset(src[d], d + 1, nbytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st[d], hipStreamNonBlocking);
    if (e != hipSuccess) { fail(e, "setting up the probe buffers", d, d); break; }
    for (int p = 0; p < n; ++p) {
      int can = 1;
      if (p != d) {
        e = hipDeviceCanAccessPeer(&can, d, p);
        if (e != hipSuccess) { fail(e, "hipDeviceCanAccessPeer", d, p); break; }
      }
      topo->can_access[d][p] = can ? 1 : 0;
    }
  }
  for (int d = 0; d < n && rc == GGMS_OK; ++d) {
    hipError_t e = hipSetDevice(d);
    for (int p = 0; p < n && e == hipSuccess; ++p)
      if (p != d && topo->can_access[d][p]) {
        e = hipDeviceEnablePeerAccess(p, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
      }
    if (e != hipSuccess) { fail(e, "hipDeviceEnablePeerAccess", d, -1); break; }
    for (int p = 0; p < n && rc == GGMS_OK; ++p) {
      if (!topo->can_access[d][p]) continue;
      double sec = 0;
      const int rc2 = timed(st[d], reps, [&]() -> int {
        // INTO device d FROM device p (the reader's view: what a gather on d would pull), :832-838
        return hipMemcpyAsync(buf[d], src[p], nbytes, hipMemcpyDefault, st[d]) == hipSuccess ? GGMS_OK : GGMS_ERR_HIP;
      }, &sec);
      if (rc2 != GGMS_OK) { fail(hipGetLastError(), "timed copy", d, p); break; }
      // the diagonal is a local copy: read + write (2 x bytes), as the reference counts it (:838-842)
      topo->copy_GBps[d][p] = (d == p ? 2.0 : 1.0) * (double)nbytes / sec / 1e9;
    }
  }
  for (int d = 0; d < n; ++d) {
    if (hipSetDevice(d) != hipSuccess) continue;
    if (st[d]) (void)hipStreamDestroy(st[d]);
    if (buf[d]) (void)hipFree(buf[d]);
    if (src[d]) (void)hipFree(src[d]);
  }
  return rc;
}

// the reference's topology file (dist_graph.cu:850-883): its loader only looks at "GPU Count", "P2P Matrix" and
// "Bandwidth Matrix" (:886-915)
int ggms_topology_write_host(const ggms_topology_t *topo, const char *path, const char *device_order) {
  GGMS_CHECK_ARG(topo && path && topo->num_device >= 0 && topo->num_device <= GGMS_TOPO_MAX_DEVICE);
  const std::string tmp = std::string(path) + ".tmp";
  FILE *f = fopen(tmp.c_str(), "w");
  if (!f) {
    set_error("ggms_topology_write_host: cannot open %s", tmp.c_str());
    return GGMS_ERR_INVALID;
  }
  const int n = topo->num_device;
  fprintf(f, "GPU Count %d\n", n);
  fprintf(f, "Device Order %s\n", device_order ? device_order : "");
  for (int i = 0; i < n; ++i) fprintf(f, "GPU [%d] gfx950\n", i);
  fprintf(f, "\n\nP2P Matrix\n");
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) fprintf(f, "%4d ", topo->can_access[i][j]);
    fprintf(f, "\n");
  }
  fprintf(f, "\n\nBandwidth Matrix\n");
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) fprintf(f, "%8.2f ", topo->copy_GBps[i][j]);
    fprintf(f, "\n");
  }
  const bool ok = fclose(f) == 0 && rename(tmp.c_str(), path) == 0; // complete or absent, never half a file
  if (!ok) {
    set_error("ggms_topology_write_host: cannot write %s", path);
    return GGMS_ERR_INVALID;
  }
  return GGMS_OK;
}

int ggms_topology_read_host(ggms_topology_t *topo, const char *path) {
  GGMS_CHECK_ARG(topo && path);
  std::memset(topo, 0, sizeof(*topo));
  FILE *f = fopen(path, "r");
  if (!f) {
    set_error("ggms_topology_read_host: cannot open %s", path);
    return GGMS_ERR_INVALID;
  }
  char line[4096];
  int n = -1;
  bool p2p = false, bw = false, bad = false;
  auto read_matrix = [&](bool ints) {
    for (int i = 0; i < n && !bad; ++i) {
      if (!fgets(line, sizeof(line), f)) { bad = true; break; }
      char *p = line;
      for (int j = 0; j < n; ++j) {
        char *end = nullptr;
        const double v = strtod(p, &end);
        if (end == p) { bad = true; break; }
        if (ints) topo->can_access[i][j] = (int32_t)v;
        else topo->copy_GBps[i][j] = v;
        p = end;
      }
    }
  };
  while (!bad && fgets(line, sizeof(line), f)) {
    if (n < 0) {
      if (sscanf(line, "GPU Count %d", &n) != 1 || n < 0 || n > GGMS_TOPO_MAX_DEVICE) bad = true; // CHECK(regex_search ...), :892
      continue;
    }
    if (strstr(line, "P2P Matrix")) { read_matrix(true); p2p = true; }
    else if (strstr(line, "Bandwidth Matrix")) { read_matrix(false); bw = true; }
  }
  fclose(f);
  if (bad || n < 0 || !p2p || !bw) {
    set_error("ggms_topology_read_host: %s is not a topology file (GPU Count / P2P Matrix / Bandwidth Matrix)", path);
    return GGMS_ERR_INVALID;
  }
  topo->num_device = n;
  return GGMS_OK;
}

// A rank's own measurement on a mapping it holds: copy `bytes` from src (local HBM, a hipIpc-mapped peer, mapped
// host memory) into dst (local HBM), `reps` times after one warm-up copy -- with hipMemcpyAsync (the copy engines / the
// runtime's blit kernel), or with_kernel != 0: a plain 16-B-per-lane streaming kernel (in-kernel loads over the link;
// locally: the device's copy ceiling).  Synchronises `stream`.
int ggms_link_probe_copy(void *dst, const void *src, size_t bytes, int reps, int with_kernel, double *GBps,
                         ggms_stream_t stream) {
  GGMS_CHECK_ARG(dst && src && bytes && GBps);
  GGMS_CHECK_ARG(!with_kernel || (bytes % 16 == 0 && (((uintptr_t)dst | (uintptr_t)src) & 15) == 0));
  if (reps < 1) reps = 1;
  hipStream_t s = to_stream(stream);
  double sec = 0;
  const int rc = timed(s, reps, [&]() -> int {
    if (with_kernel) {
      // with_kernel = workgroups per CU (256 CUs): 1 .. 16
      hipLaunchKernelGGL(k_probe_copy16, dim3(256 * (with_kernel > 16 ? 16 : with_kernel)), dim3(kBlock), 0, s,
                         (u32x4_probe *)dst, (const u32x4_probe *)src, (uint64_t)(bytes / 16));
      GGMS_LAUNCH_CHECK();
      return GGMS_OK;
    }
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s) != hipSuccess) {
      set_error("ggms_link_probe_copy: %s", hipGetErrorString(hipGetLastError()));
      return GGMS_ERR_HIP;
    }
    return GGMS_OK;
  }, &sec);
  if (rc != GGMS_OK) return rc;
  *GBps = (double)bytes / sec / 1e9;
  return GGMS_OK;
}

// The product's read pattern on the same mappings: ggms_gather_scatter_partition (the kernel behind the `peer` store's
// remote rows) gathering num_rows random rows of row_bytes from `parts` (slot s -> parts[s % num_part] row s /
// num_part, each part holding rows_per_part rows) into out[num_rows][row_bytes]; index_ws: num_rows ids of scratch.
// GBps = rows * row_bytes read per second.  One part = one peer alone; all peers = a rank's whole inbound xGMI.
// Synchronises `stream`.
int ggms_link_probe_gather(void *out, const void *const *parts, uint32_t num_part, size_t rows_per_part,
                           size_t row_bytes, size_t num_rows, uint32_t seed, int reps, ggms_id_t *index_ws,
                           double *GBps, ggms_stream_t stream) {
  GGMS_CHECK_ARG(out && parts && num_part >= 1 && rows_per_part && row_bytes && num_rows && index_ws && GBps);
  GGMS_CHECK_ARG(row_bytes % 4 == 0 && (uint64_t)rows_per_part * num_part < (1ull << 32) && num_rows < (1ull << 32));
  if (reps < 1) reps = 1;
  hipStream_t s = to_stream(stream);
  hipLaunchKernelGGL(k_probe_index, dim3(grid_for(num_rows, kBlock)), dim3(kBlock), 0, s, index_ws, (uint64_t)num_rows,
                     (uint32_t)(rows_per_part * num_part), seed);
  GGMS_LAUNCH_CHECK();
  double sec = 0;
  const int rc = timed(s, reps, [&]() -> int {
    return ggms_gather_scatter_partition(out, parts, num_part, index_ws, nullptr, num_rows, nullptr, row_bytes / 4,
                                         GGMS_I32, stream);
  }, &sec);
  if (rc != GGMS_OK) return rc;
  *GBps = (double)num_rows * (double)row_bytes / sec / 1e9;
  return GGMS_OK;
}

} // extern "C"
