// hashtable.hip -- ordered hash table: dedup + global -> local id remap.
//
// Reference: OrderedHashTable (cuda/cuda_hashtable.h:103-153, .cu:699-1064),
// DeviceOrderedHashTable::SearchO2N (.h:56-98), GPUMapEdges (cuda_mapping.cu:31-81).
//
// Same contract -- contiguous local ids, prefix-stable across fills, n2o = the
// unique list, O(1) Reset through a version stamp -- with one deliberate
// strengthening: the reference lets whichever duplicate's CAS lands first own a
// key (cuda_hashtable.cu:54-90), so the order among new ids is a race.  Here the
// FIRST occurrence in the input owns the key, always (canonical semantics, see
// DESIGN.md): ownership is decided by a 64-bit atomicMin, not by CAS arrival.
//
// Hashed bucket (16 B, two 64-bit words); the direct layout keeps only w1, one per node id:
//   w0 = { version : 32 | key : 32 }                        claimed by 64-bit CAS
//   w1 = { (0x7fffffff - version) : 31 | pending : 1 | value : 32 }
//        pending = 1: value = smallest input index seen for the key (atomicMin)
//        pending = 0: value = local id (assigned)
//   Ordering of w1 under unsigned 64-bit min does all the work:
//     newer version  <  older version   (stale buckets lose to anything current)
//     assigned       <  pending         (a key that already has a local id keeps it)
//     smaller index  <  larger index    (first occurrence wins)
//   so insertion is {CAS on w0 if needed, one atomicMin on w1}; there is no
//   per-bucket initialisation that could race with a concurrent duplicate.
// Probing is the reference's: pos = key mod size, then pos = (pos + delta++) mod size.
//
// Direct layout, who owns a key: the insert is a RETURNING atomicMin and its return value already says it
// (DedupInsert in ggms_device.h: cand / lost), so the owner scan reads no table word.  Two modes:
//   leaf  (ggms_hashtable_fill_with_duplicates / ggms_map_edges): the owner scan rewrites the owners' words as
//         {assigned, local id}, which is what SearchO2N reads and what makes later fills lose to them;
//   batch (ggms_sample_batch): one index space for the whole batch, the table is never rewritten, local ids live in
//         the fills' own outputs (IdxMap) -- k_owner_scan<true>, k_map_rest_all.
// The hashed layout keeps the reference's sizing (TableSize) and the re-read form (OwnerFlag / AssignLocal).
#include <algorithm>
#include <atomic>

#include "ggms_internal.h"
#include "tile_scan.h"

namespace ggms {

__device__ __forceinline__ unsigned long long make_w0(uint32_t version, uint32_t key) {
  return ((unsigned long long)key << 32) | version;
}
__device__ __forceinline__ uint32_t w0_version(unsigned long long w) { return (uint32_t)w; }

// Two layouts share the w1 word and all the ordering logic:
//   hashed : o2n = buckets {w0, w1} (16 B), open addressing, the reference's sizing;
//   direct : o2n = one w1 word (8 B) PER NODE ID, indexed by the id itself -- no key word, no
//            CAS, no probing: an insert is a single fire-and-forget 64-bit atomicMin.  It costs
//            8 B x num_node of HBM (products 20 MB, papers100M 0.9 GB of 288 GB) and is what the
//            engine uses; the hashed layout stays for callers that size by batch, as the
//            reference does (cuda_hashtable.cu:146-149).
struct Table {
  unsigned long long *w;
  uint32_t *n2o;
  uint32_t mask; // hashed: o2n_size - 1
  uint32_t version;

  template <bool DIRECT>
  __device__ __forceinline__ unsigned long long *w1(uint32_t pos) const {
    return DIRECT ? (w + pos) : (w + 2ull * pos + 1);
  }

  // hashed only: bucket position of `key`, inserting it if absent.  The probe sequence (triangular steps over a
  // power-of-two table) visits every bucket once in mask + 1 steps: after that the table is full -- the reference
  // would spin forever / trip its assert; here kErrTableFull goes into the status word and 0xffffffff comes back.
  __device__ __forceinline__ uint32_t find_or_claim(uint32_t key, uint32_t *err) const {
    uint32_t pos = key & mask;
    uint32_t delta = 1;
    const unsigned long long want = make_w0(version, key);
    for (;;) {
      if (delta > mask + 1u) {
        if (err) atomicOr(err, kErrTableFull);
        return 0xffffffffu;
      }
      unsigned long long *p0 = w + 2ull * pos;
      unsigned long long cur = *p0;
      if (w0_version(cur) != version) {
        const unsigned long long prev = atomicCAS(p0, cur, want);
        if (prev == cur) return pos; // claimed
        cur = prev;                  // somebody else wrote this bucket meanwhile
      }
      if (cur == want) return pos;
      if (w0_version(cur) != version) continue; // stale again (cannot happen twice, but stay safe)
      pos = (pos + delta) & mask;
      ++delta;
    }
  }

  // SearchO2N: position of a key that is present (hashed: bounded probing; direct: the id)
  template <bool DIRECT>
  __device__ __forceinline__ uint32_t find(uint32_t key) const {
    if (DIRECT) return key;
    uint32_t pos = key & mask;
    uint32_t delta = 1;
    const unsigned long long want = make_w0(version, key);
    for (uint32_t probes = 0; probes <= mask; ++probes) { // bounded: a missing key must not hang the GPU
      if (w[2ull * pos] == want) return pos;
      pos = (pos + delta) & mask;
      ++delta;
    }
    return 0xffffffffu;
  }
};

// generate_hashmap_duplicates (cuda_hashtable.cu:151-168) + ownership by atomicMin
// pro: the batch prologue rides on the first kernel of a batch (three 5-us launches less)
template <bool DIRECT>
__global__ __launch_bounds__(kBlock) void k_ht_insert(Table t, const uint32_t *__restrict__ items, Count n_arg,
                                                      uint32_t *__restrict__ item_pos, BatchPrologue pro,
                                                      DedupInsert di, uint32_t *err) {
  const uint64_t n = n_arg.get();
  if (blockIdx.x == 0) {
    for (uint32_t z = threadIdx.x; z < pro.num_zero; z += kBlock) pro.zero_words[z] = 0u;
    if (threadIdx.x == 0) {
      if (pro.num_items) *pro.num_items = 0u;
      if (pro.record_n) *pro.record_n = n;
    }
  }
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const uint32_t key = items[i];
    if (DIRECT) {
      di.enter(key, (uint32_t)i); // one returning atomicMin; cand / lost say who owns the key (ggms_device.h)
    } else {
      const uint32_t pos = t.find_or_claim(key, err);
      item_pos[i] = pos;
      if (pos != 0xffffffffu) atomicMin(t.w1<false>(pos), make_w1(t.version, 1u, (uint32_t)i));
    }
  }
}

// count_hashmap / compact_hashmap (cuda_hashtable.cu:197-232, 406-458), hashed layout:
// instance i owns its key iff the word still says {pending, i}
template <bool DIRECT>
struct OwnerFlag {
  Table t;
  const uint32_t *item_pos; // bucket positions
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
    const uint32_t pos = item_pos[i];
    if (pos == 0xffffffffu) return 0u; // table full (status word set): the item is dropped
    return *t.w1<DIRECT>(pos) == make_w1(t.version, 1u, (uint32_t)i) ? 1u : 0u;
  }
};
// mapped (optional): local id of instance i -- known right here for the owners; the others get kEmptyKey and
// are looked up by k_map_rest once every owner has written its word
template <bool DIRECT>
struct AssignLocal {
  Table t;
  const uint32_t *items;
  const uint32_t *item_pos;
  uint32_t *mapped;
  __device__ __forceinline__ void operator()(uint64_t i, uint32_t flag, uint32_t local) const {
    if (flag) {
      *t.w1<DIRECT>(item_pos[i]) = make_w1(t.version, 0u, local);
      t.n2o[local] = items[i];
    }
    if (mapped) mapped[i] = flag ? local : kEmptyKey;
  }
};

// map_edge_ids, cuda_mapping.cu:49-66
template <bool DIRECT>
__global__ __launch_bounds__(kBlock) void k_map_edges(Table t, const uint32_t *__restrict__ gsrc,
                                                      uint32_t *__restrict__ nsrc,
                                                      const uint32_t *__restrict__ gdst,
                                                      uint32_t *__restrict__ ndst, Count n_arg) {
  const uint64_t n = n_arg.get();
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    if (gsrc) {
      const uint32_t p = t.find<DIRECT>(gsrc[i]);
      nsrc[i] = (p == 0xffffffffu) ? kEmptyKey : (uint32_t)*t.w1<DIRECT>(p);
    }
    if (gdst) {
      const uint32_t p = t.find<DIRECT>(gdst[i]);
      ndst[i] = (p == 0xffffffffu) ? kEmptyKey : (uint32_t)*t.w1<DIRECT>(p);
    }
  }
}

__global__ void k_copy_prefix(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                              const uint32_t *__restrict__ count) {
  const uint64_t n = *count;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

static inline Table table_of(const ggms_hashtable_t *ht) {
  return Table{(unsigned long long *)ht->o2n, ht->n2o, (uint32_t)(ht->o2n_size - 1), ht->version};
}

// the instances that do not own their key (AssignLocal / k_owner_scan left kEmptyKey): read the owner's local id
template <bool DIRECT>
__global__ __launch_bounds__(kBlock) void k_map_rest(Table t, const uint32_t *__restrict__ item_pos, Count n_arg,
                                                     uint32_t *__restrict__ out) {
  const uint64_t n = n_arg.get();
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    if (out[i] == kEmptyKey) {
      const uint32_t pos = item_pos[i];
      if (DIRECT || pos != 0xffffffffu) out[i] = (uint32_t)*t.w1<DIRECT>(pos);
    }
}

// Batch mode, every layer in ONE launch (blockIdx.y = job): the instances that lost to another instance of their
// own fill.  A key's word is final once its fill is over, so the look-ups of all layers can wait until the last
// fill is done.  Also hands the device status word to the batch's counts (counts_dev[3 L + 1]).
__global__ __launch_bounds__(kBlock) void k_map_rest_all(const unsigned long long *__restrict__ w, MapRestJobs jobs,
                                                         IdxMap map, uint32_t *status, uint64_t *status_out) {
  // the batch TAKES the device word (exchange with 0): a failure is reported to the batch that ends next, once,
  // and does not mark every later batch of the device as failed
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && status_out) *status_out = status ? atomicExch(status, 0u) : 0u;
  const uint32_t l = blockIdx.y;
  uint32_t *__restrict__ row = jobs.row[l];
  const uint32_t *__restrict__ key = jobs.key[l];
  const uint64_t n = *jobs.num[l];
  // four consecutive entries per lane and step (one 16-byte load of `row`); the ~10 % that are still empty issue
  // their key and table reads together
  const bool vec_ok = ((((uintptr_t)row) | ((uintptr_t)key)) & 15u) == 0;
  const uint64_t n4 = vec_ok ? n / 4 : 0;
  for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < n4; q += (uint64_t)gridDim.x * kBlock) {
    const uint4 r = reinterpret_cast<const uint4 *>(row)[q];
    if (r.x != kEmptyKey && r.y != kEmptyKey && r.z != kEmptyKey && r.w != kEmptyKey) continue;
    const uint4 k = reinterpret_cast<const uint4 *>(key)[q];
    unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (r.x == kEmptyKey) w0 = w[k.x];
    if (r.y == kEmptyKey) w1 = w[k.y];
    if (r.z == kEmptyKey) w2 = w[k.z];
    if (r.w == kEmptyKey) w3 = w[k.w];
    // the word holds the INDEX of the key's owner (batch mode); the owner's entry of its fill's output is its local id
    if (r.x == kEmptyKey) row[4 * q + 0] = map.local_of((uint32_t)w0);
    if (r.y == kEmptyKey) row[4 * q + 1] = map.local_of((uint32_t)w1);
    if (r.z == kEmptyKey) row[4 * q + 2] = map.local_of((uint32_t)w2);
    if (r.w == kEmptyKey) row[4 * q + 3] = map.local_of((uint32_t)w3);
  }
  for (uint64_t i = 4 * n4 + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    if (row[i] == kEmptyKey) row[i] = map.local_of((uint32_t)w[key[i]]);
}

// ---- direct layout: owners -> local ids, one launch -----------------------------------------------------------
// count_hashmap + scan + compact_hashmap (cuda_hashtable.cu:197-232, 406-458, 812-827) as ONE single-pass ordered
// scan.  Ownership comes from the insert's own return values (DedupInsert: cand / lost), so the pass reads three
// coalesced streams and touches the table only to WRITE the owners' assigned words.  A thread owns 8 consecutive
// items (two 16-byte loads per stream), a tile is 2048 items; tiles are taken from a ticket and chained by
// decoupled look-back (tile_scan.h).  mapped[i] = local id for the owners, kEmptyKey for the rest (looked up later).
constexpr uint32_t kOwnItems = 8, kOwnTile = kOwnItems * kBlock;

template <bool BATCH>
__global__ __launch_bounds__(kBlock) void k_owner_scan(unsigned long long *__restrict__ w, uint32_t version,
                                                       uint32_t *__restrict__ n2o,
                                                       const uint32_t *__restrict__ items,
                                                       const uint32_t *__restrict__ cand,
                                                       const unsigned long long *__restrict__ lost,
                                                       unsigned long long tag, uint32_t *__restrict__ mapped,
                                                       Count n_arg, uint32_t *ctl, unsigned long long *desc,
                                                       uint32_t epoch, uint32_t *num_items, uint64_t *mirror_a,
                                                       uint64_t *mirror_b, uint32_t *err) {
  constexpr uint32_t FLAG_A = 1, FLAG_P = 2;
  __shared__ uint32_t smem[kBlock / kWave];
  __shared__ uint32_t s_tile, s_prefix;
  const uint64_t n = n_arg.get();
  const uint64_t num_tiles = (n + kOwnTile - 1) / kOwnTile;
  const uint32_t base = *num_items; // only the block that takes tile 0 uses it; the total is written last of all
  // 16-byte accesses need 16-byte aligned streams (workspace pieces are; a caller's sliced tensor may not be)
  const bool vec_ok = ((((uintptr_t)items) | ((uintptr_t)cand) | ((uintptr_t)lost) | ((uintptr_t)mapped)) & 15u) == 0;
  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&ctl[0], 1u);
    __syncthreads();
    const uint64_t tile = s_tile;
    if (tile >= num_tiles) break;
    const uint64_t i0 = tile * kOwnTile + (uint64_t)threadIdx.x * kOwnItems;
    uint32_t key[kOwnItems], flag[kOwnItems], cnd[kOwnItems];
    if (vec_ok && i0 + kOwnItems <= n) {
      const uint4 c0 = *reinterpret_cast<const uint4 *>(cand + i0), c1 = *reinterpret_cast<const uint4 *>(cand + i0 + 4);
      const uint4 k0 = *reinterpret_cast<const uint4 *>(items + i0), k1 = *reinterpret_cast<const uint4 *>(items + i0 + 4);
      const ulonglong2 l0 = *reinterpret_cast<const ulonglong2 *>(lost + i0);
      const ulonglong2 l1 = *reinterpret_cast<const ulonglong2 *>(lost + i0 + 2);
      const ulonglong2 l2 = *reinterpret_cast<const ulonglong2 *>(lost + i0 + 4);
      const ulonglong2 l3 = *reinterpret_cast<const ulonglong2 *>(lost + i0 + 6);
      key[0] = k0.x; key[1] = k0.y; key[2] = k0.z; key[3] = k0.w; key[4] = k1.x; key[5] = k1.y; key[6] = k1.z; key[7] = k1.w;
      cnd[0] = c0.x; cnd[1] = c0.y; cnd[2] = c0.z; cnd[3] = c0.w; cnd[4] = c1.x; cnd[5] = c1.y; cnd[6] = c1.z; cnd[7] = c1.w;
      flag[0] = c0.x == 1u && l0.x != tag; flag[1] = c0.y == 1u && l0.y != tag; flag[2] = c0.z == 1u && l1.x != tag;
      flag[3] = c0.w == 1u && l1.y != tag; flag[4] = c1.x == 1u && l2.x != tag; flag[5] = c1.y == 1u && l2.y != tag;
      flag[6] = c1.z == 1u && l3.x != tag; flag[7] = c1.w == 1u && l3.y != tag;
    } else {
#pragma unroll
      for (uint32_t k = 0; k < kOwnItems; ++k) {
        const uint64_t i = i0 + k;
        key[k] = i < n ? items[i] : 0u;
        cnd[k] = i < n ? cand[i] : 0u;
        flag[k] = (i < n && cnd[k] == 1u && lost[i] != tag) ? 1u : 0u;
      }
    }
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kOwnItems; ++k) mine += flag[k];
    uint32_t running;
    const uint32_t excl = block_exclusive_scan(mine, smem, running);
    if (threadIdx.x < kWave) { // wave 0 publishes the tile and looks back
      const uint32_t lane = threadIdx.x;
      uint32_t prefix = base;
      if (tile == 0) {
        if (lane == 0)
          __hip_atomic_store(&desc[0], scan_desc(epoch, FLAG_P, base + running), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (lane == 0)
          __hip_atomic_store(&desc[tile], scan_desc(epoch, FLAG_A, running), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prefix = scan_lookback(desc, tile, epoch, err);
        if (lane == 0)
          __hip_atomic_store(&desc[tile], scan_desc(epoch, FLAG_P, prefix + running), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        s_prefix = prefix;
        if (tile + 1 == num_tiles) {
          const uint32_t total = prefix + running;
          *num_items = total;
          if (mirror_a) *mirror_a = (uint64_t)total;
          if (mirror_b) *mirror_b = (uint64_t)total;
        }
      }
    }
    __syncthreads();
    uint32_t local = s_prefix + excl;
    uint32_t out[kOwnItems];
#pragma unroll
    for (uint32_t k = 0; k < kOwnItems; ++k) {
      // batch mode: an instance that lost to an earlier fill carries its local id in cand (2 + id), and the
      // table is not rewritten (the owner's index stays there: IdxMap); leaf mode: the owner's word becomes
      // {assigned, local id}
      out[k] = (BATCH && cnd[k] >= 2u) ? cnd[k] - 2u : kEmptyKey;
      if (flag[k]) {
        if (!BATCH) w[key[k]] = make_w1(version, 0u, local);
        n2o[local] = key[k];
        out[k] = local++;
      }
    }
    if (mapped) {
      if (vec_ok && i0 + kOwnItems <= n) {
        *reinterpret_cast<uint4 *>(mapped + i0) = make_uint4(out[0], out[1], out[2], out[3]);
        *reinterpret_cast<uint4 *>(mapped + i0 + 4) = make_uint4(out[4], out[5], out[6], out[7]);
      } else {
#pragma unroll
        for (uint32_t k = 0; k < kOwnItems; ++k)
          if (i0 + k < n) mapped[i0 + k] = out[k];
      }
    }
    __syncthreads(); // s_tile / s_prefix are rewritten next iteration
  }
  if (threadIdx.x == 0) {
    if (num_tiles == 0 && blockIdx.x == 0) { // empty input: the count stays what it is
      if (mirror_a) *mirror_a = (uint64_t)base;
      if (mirror_b) *mirror_b = (uint64_t)base;
    }
    if (atomicAdd(&ctl[1], 1u) == gridDim.x - 1) { // the last block out re-arms the control words
      ctl[0] = 0;
      ctl[1] = 0;
    }
  }
}

static std::atomic<bool> g_poison_next_scan{false};

static size_t owner_scan_grid_cap() { // GGMS_OSCAN_GRID: measurement hook
  static const size_t v = [] { const char *e = getenv("GGMS_OSCAN_GRID"); const long x = e ? atol(e) : 0; return x > 0 ? (size_t)x : (size_t)512; }();
  return v;
}

// descriptors the owner scan needs for n_max items (64-bit words behind the 8 control words)
size_t owner_scan_tiles(size_t n_max) { return (n_max + kOwnTile - 1) / kOwnTile; }

// workspace of one fill: cand / item_pos [n], lost [n] (64-bit), scan area
size_t ht_ws_words(size_t num_input) { return 3 * num_input + tile_scan_words(num_input) + 24; }

__global__ void k_status_copy(uint32_t *status, uint64_t *status_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *status_out = status ? atomicExch(status, 0u) : 0u;
}

int launch_map_rest_all(const ggms_hashtable_t *ht, const MapRestJobs &jobs, uint32_t num_jobs, size_t max_items,
                        const IdxMap &map, uint64_t *status_out, hipStream_t s) {
  if (num_jobs == 0) { // nothing deferred (hashed layout, or no layer could sample): only the status word
    if (!status_out) return GGMS_OK;
    hipLaunchKernelGGL(k_status_copy, dim3(1), dim3(64), 0, s, device_status_word(), status_out);
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  const int gx = grid_for(max_items ? max_items : 1, kBlock);
  hipLaunchKernelGGL(k_map_rest_all, dim3(gx, num_jobs), dim3(kBlock), 0, s, (const unsigned long long *)ht->o2n, jobs, map,
                     status_out ? device_status_word() : (uint32_t *)nullptr, status_out);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

// mapped != NULL: also produce the local id of every input instance (FillWithDuplicates + the dst half of
// GPUMapEdges in one go): owners write theirs while assigning, the rest is looked up afterwards.
// prologue: this is the first kernel of a batch -- it also zeroes the shared scan area's control words.
// Direct layout: `di` carries cand / lost (and, in batch mode, the index base + IdxMap of the earlier fills);
// inserted = the producer of `input` already entered every item with exactly this `di` (fused sampler).
// rest: kRestNow = resolve the non-owners here; kRestDefer (batch mode) = the caller's launch_map_rest_all does.
int ht_fill_impl(const ggms_hashtable_t *ht, const uint32_t *input, size_t n_max, Count n, DedupInsert di,
                 bool inserted, ScanArea scratch, uint64_t *mirror_a, uint64_t *mirror_b, hipStream_t s,
                 uint32_t *mapped, const BatchPrologue *prologue, int rest, const uint64_t *n_dev_for_rest) {
  BatchPrologue pro{nullptr, 0, nullptr, nullptr};
  if (prologue) pro = *prologue;
  if (n_max == 0) { // no kernel to ride on
    if (pro.num_zero) GGMS_HIP(hipMemsetAsync(pro.zero_words, 0, pro.num_zero * sizeof(uint32_t), s));
    if (pro.num_items) GGMS_HIP(hipMemsetAsync(pro.num_items, 0, sizeof(uint32_t), s));
    if (pro.record_n) GGMS_HIP(hipMemsetAsync(pro.record_n, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  Table t = table_of(ht);
  uint32_t *err = device_status_word();
  const int grid = grid_for(n_max, kBlock);
  uint32_t *item_pos = di.cand; // hashed layout: bucket positions
  if (!ht->direct) {
    hipLaunchKernelGGL(k_ht_insert<false>, dim3(grid), dim3(kBlock), 0, s, t, input, n, item_pos, pro, di, err);
    GGMS_LAUNCH_CHECK();
    int rc = tile_scan(OwnerFlag<false>{t, item_pos}, AssignLocal<false>{t, input, item_pos, mapped}, n_max, n, scratch,
                       ht->num_items_dev, ht->num_items_dev, nullptr, s, mirror_a, mirror_b);
    if (rc != GGMS_OK || !mapped) return rc;
    hipLaunchKernelGGL(k_map_rest<false>, dim3(grid), dim3(kBlock), 0, s, t, item_pos, n, mapped);
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  di.w = (unsigned long long *)ht->o2n;
  di.version = ht->version;
  if (!inserted) {
    di.tag = next_dedup_tag();
    hipLaunchKernelGGL(k_ht_insert<true>, dim3(grid), dim3(kBlock), 0, s, t, input, n, item_pos, pro, di, err);
    GGMS_LAUNCH_CHECK();
  }
  uint32_t *ctl = scan_align(scratch.words);
  unsigned long long *desc = reinterpret_cast<unsigned long long *>(ctl + 8);
  if (!scratch.cleared) {
    GGMS_HIP(hipMemsetAsync(ctl, 0, (8 + 2 * (owner_scan_tiles(n_max) + 1)) * sizeof(uint32_t), s));
    // ggms_debug_poison_next_scan (tests): start the ticket at 1, so tile 0 is never processed and every later
    // tile's look-back runs into its bound -- the failure the status word exists for
    if (g_poison_next_scan.exchange(false)) GGMS_HIP(hipMemsetD32Async((hipDeviceptr_t)ctl, 1, 1, s));
  }
  // Fewer workgroups than tiles on purpose: a workgroup takes tiles from the ticket one after the other, so by the
  // time tile t is taken the tiles before t - grid have finished and the look-back finds a published prefix in
  // its first window; with one workgroup per tile every tile starts at once and tile t walks t / 64 windows.
  const int oscan_grid = (int)std::min<size_t>(grid_for(owner_scan_tiles(n_max), 1), owner_scan_grid_cap());
  if (di.batch)
    hipLaunchKernelGGL(k_owner_scan<true>, dim3(oscan_grid), dim3(kBlock), 0, s, di.w, di.version, ht->n2o, input, di.cand,
                       di.lost, di.tag, mapped, n, ctl, desc, next_scan_epoch(), ht->num_items_dev, mirror_a, mirror_b, err);
  else
    hipLaunchKernelGGL(k_owner_scan<false>, dim3(oscan_grid), dim3(kBlock), 0, s, di.w, di.version, ht->n2o, input, di.cand,
                       di.lost, di.tag, mapped, n, ctl, desc, next_scan_epoch(), ht->num_items_dev, mirror_a, mirror_b, err);
  GGMS_LAUNCH_CHECK();
  if (!mapped || rest == kRestDefer) return GGMS_OK;
  if (di.batch) { // this fill's own non-owners, now: the map must already hold this fill's segment
    MapRestJobs jobs{};
    jobs.row[0] = mapped;
    jobs.key[0] = input;
    jobs.num[0] = n_dev_for_rest;
    return launch_map_rest_all(ht, jobs, 1, n_max, di.map, nullptr, s);
  }
  hipLaunchKernelGGL(k_map_rest<true>, dim3(grid), dim3(kBlock), 0, s, t, input, n, mapped);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // namespace ggms

using namespace ggms;

extern "C" {

// TableSize(num, scale = kDefaultScale = 2): cuda_hashtable.cu:146-149, cuda_hashtable.h:105
void ggms_debug_poison_next_scan(void) { g_poison_next_scan.store(true); }

size_t ggms_hashtable_num_buckets(size_t capacity) {
  size_t half = capacity >> 1;
  size_t lg = 0; // floor(log2(half)); the reference's std::log2(0) = -inf case maps to 1 bucket << scale
  while ((half >> (lg + 1)) != 0) ++lg;
  const size_t next_pow2 = half == 0 ? 1 : ((size_t)1 << (1 + lg));
  return next_pow2 << 2;
}

int ggms_hashtable_init(ggms_hashtable_t *ht, ggms_stream_t stream) {
  GGMS_CHECK_ARG(ht && ht->o2n && ht->n2o && ht->num_items_dev);
  GGMS_CHECK_ARG(ht->o2n_size != 0 && ht->o2n_size <= (1ull << 32));
  GGMS_CHECK_ARG(ht->direct || (ht->o2n_size & (ht->o2n_size - 1)) == 0);
  GGMS_HIP(hipMemsetAsync(ht->o2n, 0xff, ht->o2n_size * (ht->direct ? 8 : 16), to_stream(stream)));
  GGMS_HIP(hipMemsetAsync(ht->n2o, 0xff, ht->n2o_size * sizeof(uint32_t), to_stream(stream)));
  GGMS_HIP(hipMemsetAsync(ht->num_items_dev, 0, sizeof(uint32_t), to_stream(stream)));
  ht->version = 0;
  return GGMS_OK;
}

int ggms_hashtable_reset(ggms_hashtable_t *ht, ggms_stream_t stream) {
  GGMS_CHECK_ARG(ht && ht->num_items_dev);
  if (ht->version >= 0x7ffffff0u) {
    // version space exhausted (2^31 batches): start over with clean buckets
    int rc = ggms_hashtable_init(ht, stream);
    if (rc != GGMS_OK) return rc;
  }
  ht->version += 1;
  GGMS_HIP(hipMemsetAsync(ht->num_items_dev, 0, sizeof(uint32_t), to_stream(stream)));
  return GGMS_OK;
}

size_t ggms_hashtable_workspace_bytes(size_t num_input) {
  // item_pos / cand [num_input] + lost [num_input] (64-bit) + scan scratch
  return ht_ws_words(num_input) * sizeof(uint32_t);
}

int ggms_hashtable_fill_with_duplicates(ggms_hashtable_t *ht, const ggms_id_t *input, size_t num_input,
                                        ggms_id_t *unique_out, void *workspace, size_t workspace_bytes,
                                        ggms_stream_t stream) {
  GGMS_CHECK_ARG(ht && ht->o2n && ht->version != 0);
  hipStream_t s = to_stream(stream);
  if (num_input != 0) {
    GGMS_CHECK_ARG(input && workspace);
    GGMS_CHECK_ARG(workspace_bytes >= ggms_hashtable_workspace_bytes(num_input));
    GGMS_CHECK_ARG(num_input < (1ull << 32));
    uint32_t *item_pos = (uint32_t *)workspace;
    unsigned long long *lost = (unsigned long long *)(((uintptr_t)(item_pos + num_input) + 15) & ~(uintptr_t)15);
    uint32_t *scan_words = (uint32_t *)(lost + num_input);
    DedupInsert di{};
    di.cand = item_pos;
    di.lost = lost; // leaf mode: base 0, the owners' words are rewritten as {assigned, local id}
    int rc = ht_fill_impl(ht, input, num_input, count_of(num_input), di, false, ScanArea{scan_words, false}, nullptr,
                          nullptr, s, nullptr, nullptr, kRestNow, nullptr);
    if (rc != GGMS_OK) return rc;
  }
  if (unique_out) {
    hipLaunchKernelGGL(k_copy_prefix, dim3(256), dim3(kBlock), 0, s, ht->n2o, unique_out, ht->num_items_dev);
    GGMS_LAUNCH_CHECK();
  }
  return GGMS_OK;
}

int ggms_map_edges(const ggms_hashtable_t *ht, const ggms_id_t *global_src, ggms_id_t *new_src,
                   const ggms_id_t *global_dst, ggms_id_t *new_dst, size_t num_edges, ggms_stream_t stream) {
  GGMS_CHECK_ARG(ht && ht->o2n && ht->version != 0);
  if (num_edges == 0) return GGMS_OK;
  GGMS_CHECK_ARG((global_src == nullptr) == (new_src == nullptr));
  GGMS_CHECK_ARG((global_dst == nullptr) == (new_dst == nullptr));
  const int grid = grid_for(num_edges, kBlock);
  if (ht->direct)
    hipLaunchKernelGGL(k_map_edges<true>, dim3(grid), dim3(kBlock), 0, to_stream(stream), table_of(ht), global_src,
                       new_src, global_dst, new_dst, count_of(num_edges));
  else
    hipLaunchKernelGGL(k_map_edges<false>, dim3(grid), dim3(kBlock), 0, to_stream(stream), table_of(ht), global_src,
                       new_src, global_dst, new_dst, count_of(num_edges));
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // extern "C"
