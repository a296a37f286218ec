// ggms_device.h -- shared device-side building blocks (gfx950, wave64).
#pragma once
#include <cstdlib>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ggms.h"

namespace ggms {

constexpr uint32_t kEmptyKey = GGMS_EMPTY_KEY;
constexpr int kWave = 64;
constexpr int kBlock = 256;          // 4 waves, one per SIMD of a CU
constexpr int kMaxGridBlocks = 2048; // 256 CUs x 8 resident blocks: grid-stride above

// ---- error plumbing (host) -------------------------------------------------
void set_error(const char *fmt, ...);
#define GGMS_CHECK_ARG(cond)                                              \
  do {                                                                    \
    if (!(cond)) {                                                        \
      ::ggms::set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
      return GGMS_ERR_INVALID;                                            \
    }                                                                     \
  } while (0)
#define GGMS_HIP(call)                                                    \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) {                                               \
      ::ggms::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return GGMS_ERR_HIP;                                                \
    }                                                                     \
  } while (0)
#define GGMS_LAUNCH_CHECK() GGMS_HIP(hipGetLastError())

inline hipStream_t to_stream(ggms_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Grid for a grid-stride kernel over n items, capped at every wave slot of the device (256 CUs x 32 waves =
// 2048 blocks x 256 threads).
inline size_t grid_cap() { return (size_t)kMaxGridBlocks; }
inline int grid_for(size_t n, size_t per_block) {
  size_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > grid_cap()) g = grid_cap();
  return (int)g;
}

// ---- status words ---------------------------------------------------------------
// A kernel that hits a bound it must not hit ORs a bit into a status word instead of hanging or carrying on
// silently.  Leaf operators use the sticky word of the device (allocated on first use, read by
// ggms_device_status()); the kernels of a ggms_sample_batch call use the BATCH's own word -- the one behind its
// table's item counter -- which the batch's last kernel moves into counts_dev[3 L + 1] (include/ggms.h).  The
// reference CHECK-aborts in these places (logging.cc:69-73); the engine does the same once it has seen the word.
constexpr uint32_t kErrScanSpin = 1u;    // decoupled look-back gave up waiting for a predecessor tile
constexpr uint32_t kErrTableFull = 2u;   // hashed dedup table: probing found no free bucket
uint32_t *device_status_word();          // common.hip; NULL if it cannot be allocated
long long debug_knob(int knob);          // common.hip: ggms_debug_set_knob's value, -1 = default

// ---- a count that lives either in an argument or in device memory ---------
// Lets a whole mini-batch be enqueued without a host round trip: the size of
// layer l's frontier is only known on the device when layer l+1 is enqueued.
struct Count {
  const uint64_t *dev64; // if non-null, the value is read on the device (64-bit counter)
  const uint32_t *dev32; // same for a 32-bit counter
  uint64_t imm;
  __device__ __forceinline__ uint64_t get() const {
    return dev64 ? *dev64 : (dev32 ? (uint64_t)*dev32 : imm);
  }
};
inline Count count_of(size_t n, const uint64_t *dev = nullptr) { return Count{dev, nullptr, (uint64_t)n}; }
inline Count count_of32(size_t n, const uint32_t *dev) { return Count{nullptr, dev, (uint64_t)n}; }

// ---- v / d and v % d by a launch constant ------------------------------------------------------------------
// q' = mulhi(v, floor(2^32 / d)) is q or q - 1 (v * (2^32 - m d) / (d 2^32) < v / 2^32 < 1): one fix-up.  d = 1 takes
// m = 2^32 - 1 (q' = v - 1 for v >= 1, fixed up to v).  A 32-bit division by a runtime value is ~25 instructions on
// this ISA (v_rcp_f32 + Newton step + fix-ups); shard counts are launch constants.
struct Divisor {
  uint32_t d, magic;
  __device__ __forceinline__ void divmod(uint32_t v, uint32_t &q, uint32_t &r) const {
    q = __umulhi(v, magic);
    r = v - q * d;
    const bool up = r >= d;
    q += up ? 1u : 0u;
    r -= up ? d : 0u;
  }
};
inline Divisor divisor_of(uint32_t d) {
  return Divisor{d, d <= 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / d)};
}

// ---- shard base pointers, by value ------------------------------------------------------------------------
// The pointers of up to GGMS_MAX_PARTS shards (+ one host slot) travel in the kernel arguments: a lane picks its
// shard's pointer with a chain of selects on registers instead of a dependent load from a device-side pointer table
// (the reference's DeviceDistGraph / DeviceDistFeature read `part_indptr[part]` from global memory before the first
// useful request of every lookup can be issued, dist_graph.h:150-157,196-204).
constexpr uint32_t kMaxParts = GGMS_MAX_PARTS;
template <typename T, uint32_t N>
struct PtrSet {
  T *p[N];
  __device__ __forceinline__ T *pick(uint32_t k) const {
    T *r = p[0];
#pragma unroll 1 // one scalar load + one select per round: the N pointers need not sit in SGPRs all at once
    for (uint32_t i = 1; i < N; ++i) r = (k == i) ? p[i] : r;
    return r;
  }
};

// ---- graph views (DeviceNormalGraph / DeviceDistGraph) ---------------------
struct GraphView {
  const uint32_t *indptr;
  const uint32_t *indices;
  PtrSet<const uint32_t, kMaxParts + 1> pip, pix; // shard p at [p]; the whole CSR (host tier) ALWAYS at [kMaxParts]
  Divisor part;                                   // d = num_part
  uint32_t num_part;
  uint32_t num_cache_node;

  // indptr[v], indptr[v + 1] with ONE 8-byte load (4-byte aligned: global_load_dwordx2 only needs dword alignment):
  // one request at the memory side instead of two -- the sampler is bounded by requests, not bytes
  typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
  static __device__ __forceinline__ void bounds(const uint32_t *ip, uint32_t v, uint32_t &b, uint32_t &e) {
    const u32x2_a4 be = *reinterpret_cast<const u32x2_a4 *>(ip + v);
    b = be.x;
    e = be.y;
  }
  // neighbour list of v: pointer + length
  __device__ __forceinline__ const uint32_t *neighbours(uint32_t v, uint32_t &len) const {
    if (num_part == 0) {
      uint32_t b, e;
      bounds(indptr, v, b, e);
      len = e - b;
      return indices + b;
    }
    // DeviceDistGraph::operator[] / NumEdge (dist_graph.h:132-158): v < num_cache_node -> shard v % P, row v / P;
    // every other node in the whole CSR of the last slot
    uint32_t slot, real;
    part.divmod(v, real, slot);
    const bool cached = v < num_cache_node;
    slot = cached ? slot : kMaxParts;
    real = cached ? real : v;
    uint32_t b, e;
    bounds(pip.pick(slot), real, b, e);
    len = e - b;
    return pix.pick(slot) + b;
  }
};

// The shard pointer tables of the ABI are HOST arrays since ABI 3 (they were device arrays before): a caller that still
// hands over a device array would have this library dereference device memory on the host.  One attribute query per
// call (host side) turns that into an argument error.  common.hip
bool host_readable_table(const void *table, const char *what);

// host side: NULL + an error message if the graph has more shards than the kernels carry
inline bool view_of(const ggms_graph_t *g, GraphView &v) {
  v = GraphView{};
  v.indptr = g->indptr;
  v.indices = g->indices;
  v.num_part = g->num_part;
  v.num_cache_node = g->num_cache_node;
  v.part = divisor_of(g->num_part ? g->num_part : 1);
  if (g->num_part == 0) return true;
  if (g->num_part > kMaxParts || !g->part_indptr || !g->part_indices) {
    set_error("graph view: num_part %u (at most %u shards; part_indptr / part_indices are HOST arrays of num_part + 1 device pointers)",
              g->num_part, kMaxParts);
    return false;
  }
  if (!host_readable_table(g->part_indptr, "ggms_graph_t.part_indptr") ||
      !host_readable_table(g->part_indices, "ggms_graph_t.part_indices"))
    return false;
  for (uint32_t p = 0; p < g->num_part; ++p) {
    v.pip.p[p] = g->part_indptr[p];
    v.pix.p[p] = g->part_indices[p];
  }
  v.pip.p[kMaxParts] = g->part_indptr[g->num_part];
  v.pix.p[kMaxParts] = g->part_indices[g->num_part];
  return true;
}

// ---- XORWOW, bit-compatible with cuRAND's curand_init(seed,0,0)/curand() ---
struct Xorwow {
  uint32_t d, v0, v1, v2, v3, v4;
  __host__ __device__ __forceinline__ void init(uint64_t seed) {
    const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    const uint32_t s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    d = 6615241u + t1 + t0;
    v0 = 123456789u + t0;
    v1 = 362436069u ^ t0;
    v2 = 521288629u + t1;
    v3 = 88675123u ^ t1;
    v4 = 5783321u + t0;
  }
  __host__ __device__ __forceinline__ uint32_t next() {
    const uint32_t t = v0 ^ (v0 >> 2);
    v0 = v1; v1 = v2; v2 = v3; v3 = v4;
    v4 = (v4 ^ (v4 << 4)) ^ (t ^ (t << 1));
    d += 362437u;
    return v4 + d;
  }
  // curand_uniform: x * 2^-32 + 2^-33 in f32
  __device__ static __forceinline__ float to_uniform(uint32_t x) {
    return __fmaf_rn(__uint2float_rn(x), 2.3283064e-10f, 2.3283064e-10f / 2.0f);
  }
  __device__ __forceinline__ float uniform() { return to_uniform(next()); }
  // curand_uniform_double (XORWOW): two draws, 53-bit mantissa
  __device__ __forceinline__ double uniform_double() {
    const uint32_t x = next();
    const uint32_t y = next();
    const uint64_t z = (uint64_t)x ^ ((uint64_t)y << 21);
    return __fma_rn(__ull2double_rn(z), 1.1102230246251565e-16, 1.1102230246251565e-16 / 2.0);
  }
  __device__ __forceinline__ void load(const uint32_t *p) {
    d = p[0]; v0 = p[1]; v1 = p[2]; v2 = p[3]; v3 = p[4]; v4 = p[5];
  }
  __device__ __forceinline__ void store(uint32_t *p) const {
    p[0] = d; p[1] = v0; p[2] = v1; p[3] = v2; p[4] = v3; p[5] = v4;
  }
};

// ---- ordered dedup word (direct-mapped table, one per node id; hashtable.hip has the full story) ---------
//   { (0x7fffffff - version) : 31 | pending : 1 | value : 32 } under unsigned 64-bit min:
//   newer version < older, assigned < pending, smaller first index < larger.
__host__ __device__ __forceinline__ unsigned long long make_w1(uint32_t version, uint32_t pending, uint32_t value) {
  const unsigned long long hi = ((unsigned long long)(0x7fffffffu - version) << 1) | pending;
  return (hi << 32) | value;
}

// Who owns a key without re-reading the table: instance e enters {pending, base + e} with ONE returning atomicMin.
//   old < mine : an earlier instance or an assigned id is there  -> e cannot own the key   (cand[e] = 0)
//   old > mine : e is the smallest so far                         -> candidate              (cand[e] = 1);
//                if old is a pending word of THIS fill, its instance has just been beaten: the thief says so in
//                lost[that instance] (each candidate is beaten at most once -- by the next smaller arrival -- and
//                only thieves write `lost`, only e writes cand[e]: no word has two writers).
// After the fill (kernel boundary): e owns its key  <=>  cand[e] == 1 && lost[e] != tag.  `tag` is a 64-bit launch
// counter started from a random nonce, so `lost` never needs clearing (a stale or foreign word cannot match).
//
// Two modes.  LEAF (the table API, ggms_hashtable_*): base = 0, the owner scan then rewrites the word as
// {assigned, local id} -- what SearchO2N / ggms_map_edges read, and what makes later fills lose to it.
// BATCH (ggms_sample_batch): every fill of the batch draws its indices from ONE index space -- seeds [0, S), then
// the layers in processing order -- so an earlier fill's word is simply SMALLER than anything a later fill
// offers and the table is never rewritten (one random 8-byte store per unique node saved).  The local id of the
// instance with index idx lives where the owner scan put it: IdxMap::local_of(idx) = that fill's output array.
// An instance that loses to an EARLIER fill knows its local id at once (cand[e] = 2 + id); one that loses inside
// its own fill is resolved at the end of the batch (table word -> index -> IdxMap).
struct IdxMap {
  uint32_t base[17];        // first index of segment s, ascending; segment 0 = the seeds
  const uint32_t *arr[17];  // local ids of segment s, by index - base[s] (seed_local / the layer's row);
                            // arr[0] == NULL: the seeds are distinct, local id = position
  uint32_t n;
  __device__ __forceinline__ uint32_t local_of(uint32_t idx) const {
    // n - 1 <= L rounds (uniform trip count) of scalar loads from the kernel arguments + per-lane selects: the 34 words
    // of the two arrays are not all kept in SGPRs (the fused sampler spills SGPRs as it is)
    uint32_t b = base[0];
    const uint32_t *a = arr[0];
#pragma unroll 1
    for (uint32_t k = 1; k < n; ++k) {
      const bool ge = idx >= base[k];
      b = ge ? base[k] : b;
      a = ge ? arr[k] : a;
    }
    const uint32_t off = idx - b;
    return a ? a[off] : off;
  }
};

struct DedupInsert {
  unsigned long long *w; // table words, indexed by node id
  uint32_t version;
  uint32_t *cand;
  unsigned long long *lost;
  unsigned long long tag;
  uint32_t base;  // index of this fill's item 0 (0 in leaf mode)
  uint32_t batch; // 1: batch mode, `map` holds the fills before this one
  IdxMap map;
  // enter = issue + finish: a lane with several independent instances issues them all before finishing any, so
  // that the atomics' round trips overlap
  __device__ __forceinline__ unsigned long long issue(uint32_t key, uint32_t e) const {
    return atomicMin(w + key, make_w1(version, 1u, base + e));
  }
  __device__ __forceinline__ void enter(uint32_t key, uint32_t e) const { finish(issue(key, e), e); }
  __device__ __forceinline__ void finish(unsigned long long old, uint32_t e) const {
    const unsigned long long mine = make_w1(version, 1u, base + e);
    if (old > mine) {
      cand[e] = 1u;
      if ((old >> 32) == (mine >> 32)) lost[(uint32_t)old - base] = tag; // a pending word of this fill: beaten
    } else {
      const uint32_t idx = (uint32_t)old;
      cand[e] = (batch && idx < base) ? 2u + map.local_of(idx) : 0u;
    }
  }
};
unsigned long long next_dedup_tag(); // common.hip
size_t device_lds_bytes();                                                       // common.hip
int raise_dynamic_lds(const void *func, size_t bytes, const char *who);          // common.hip

// ---- wave / block prefix sums (wave64) -------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if ((int)lane_id() >= off) x += y;
  }
  return x;
}

__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

// Exclusive prefix sum over the 256 threads of a block; `total` gets the block sum.
// smem: at least 4 uint32.  Ends with a barrier so smem can be reused.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t x, uint32_t *smem, uint32_t &total) {
  const uint32_t incl = wave_inclusive_scan(x);
  const uint32_t w = threadIdx.x >> 6;
  if (lane_id() == 63) smem[w] = incl;
  __syncthreads();
  uint32_t base = 0, t = 0;
#pragma unroll
  for (uint32_t i = 0; i < kBlock / kWave; ++i) {
    const uint32_t s = smem[i];
    if (i < w) base += s;
    t += s;
  }
  total = t;
  __syncthreads();
  return base + incl - x;
}

} // namespace ggms
