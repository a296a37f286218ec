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
  if (blockIdx.x == 0) pro.run(n, threadIdx.x, kBlock);
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

// Distinct seeds (ggms_sample_extra_t.seeds_distinct), direct layout: FillWithDupRevised(seeds) (dist_loops.cc:105-111)
// has nothing to decide -- seed i is item i of the unique list and owns its key.  ONE launch instead of insert +
// ordered scan + look-up: the table entry (index i: smaller than anything a later fill of the batch offers), the
// head of n2o, the batch prologue (item count = |seeds|).
__global__ __launch_bounds__(kBlock) void k_seed_enter(unsigned long long *__restrict__ w, uint32_t version,
                                                       const uint32_t *__restrict__ seeds, Count n_arg,
                                                       uint32_t *__restrict__ n2o, BatchPrologue pro) {
  const uint64_t n = n_arg.get();
  if (blockIdx.x == 0) pro.run(n, threadIdx.x, kBlock);
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const uint32_t key = seeds[i];
    n2o[i] = key;
    atomicMin(w + key, make_w1(version, 1u, (uint32_t)i)); // not returning: nothing of this batch was entered before
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

// the last kernel of a batch moves the BATCH's status word (the word behind its table's item counter; every kernel
// of the batch reports into that word only) into the batch's counts: exchange with 0, so the next batch on this
// table starts clean; a failure is also recorded in the device's sticky word (ggms_device_status)
__device__ __forceinline__ void take_batch_status(uint32_t *status, uint32_t *device_word, uint64_t *status_out) {
  const uint32_t v = status ? atomicExch(status, 0u) : 0u;
  *status_out = v;
  if (v && device_word) atomicOr(device_word, v);
}

// Batch mode, every layer in ONE launch (blockIdx.y = job): the instances that lost to another instance of their
// own fill.  A key's word is final once its fill is over, so the look-ups of all layers can wait until the last
// fill is done.  Also hands the batch's status word to the batch's counts (counts_dev[3 L + 1]).
__global__ __launch_bounds__(kBlock) void k_map_rest_all(const unsigned long long *__restrict__ w, MapRestJobs jobs,
                                                         IdxMap map, uint32_t *status, uint32_t *device_word,
                                                         uint64_t *status_out) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && status_out) take_batch_status(status, device_word, status_out);
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

// ---- direct layout, the ordered scan proper: chunked, request-efficient ----------------------------------------
// Two things k_owner_scan above pays for, both at the memory side where this path is bounded (tools/micro_atomics2.hip):
//   * a thread's 8 consecutive items are 32 / 64 bytes apart from its neighbour's, so every 16-byte load instruction
//     of a wave touches 32-64 different lines and each line is requested by 2-4 different instructions: three times
//     the line requests the bytes need.  Here a wave's load instruction covers ONE contiguous span: in round q of a
//     tile, thread t holds items 512 q + 2 t and 512 q + 2 t + 1 (8-byte loads of cand / items, a 16-byte load of lost);
//   * one ticket per tile from ONE counter: same-address atomics are served one at a time, 11 ns each
//     (tools/micro_ticket.hip) -- 1220 tickets, as many failing ones and exit counts are most of the old kernel's 42 us;
//     and tiles chained by decoupled look-back walk t / 64 windows of descriptors, one dependent round trip each, when
//     every workgroup starts at once.  Here workgroup c owns CHUNK c of `per` consecutive tiles (no ticket), publishes
//     the chunk's owner count (<= 16 bits: a 32-bit descriptor {epoch : 16 | count : 16}) and fetches the counts of ALL
//     chunks before it with loads that are in flight together.  Chunk 0's descriptor is 64-bit and carries the
//     table's item count in.
// Ranks come from ballots: a (tile, round, wave) segment is 128 consecutive items, its owners are counted with two
// ballots, the segment totals of the whole chunk meet in LDS after ONE barrier.  Owners park their keys in LDS at
// their rank and the chunk's slice of n2o is written in one coalesced sweep.  Chunks of up to kChunkRegTiles tiles
// stay in registers between the count and the write-out; longer ones (input near its worst-case bound) re-read.
constexpr uint32_t kChunkGrid = 1024, kChunkRegTiles = 2;
constexpr uint32_t kChunkMaxTiles = 65535u / kOwnTile; // 31: a chunk's count must fit the descriptor's 16 bits
constexpr uint32_t kOwnRounds = kOwnTile / (2 * kBlock); // 4 rounds of 512 items
constexpr uint32_t kOwnSegs = kOwnRounds * (kBlock / kWave); // 16 segments of 128 items per tile

struct OwnTile { // one thread's items of a tile: entry 2 q + j = item 512 q + 2 t + j
  uint32_t key[kOwnItems], cnd[kOwnItems], flags; // flags: bit (2 q + j) = that item owns its key
};

__device__ __forceinline__ void own_tile_load(OwnTile &t, const uint32_t *__restrict__ items,
                                              const uint32_t *__restrict__ cand,
                                              const unsigned long long *__restrict__ lost, unsigned long long tag,
                                              uint64_t tile0, uint64_t n, bool vec_ok, bool need_keys) {
  t.flags = 0;
#pragma unroll
  for (uint32_t q = 0; q < kOwnRounds; ++q) {
    const uint64_t i = tile0 + 2ull * kBlock * q + 2ull * threadIdx.x;
    uint32_t c0 = 0, c1 = 0, k0 = 0, k1 = 0;
    unsigned long long l0 = tag, l1 = tag;
    if (vec_ok && i + 2 <= n) {
      const uint2 c = *reinterpret_cast<const uint2 *>(cand + i);
      const ulonglong2 l = *reinterpret_cast<const ulonglong2 *>(lost + i);
      if (need_keys) {
        const uint2 k = *reinterpret_cast<const uint2 *>(items + i);
        k0 = k.x; k1 = k.y;
      }
      c0 = c.x; c1 = c.y; l0 = l.x; l1 = l.y;
    } else {
      if (i < n) { c0 = cand[i]; l0 = lost[i]; if (need_keys) k0 = items[i]; }
      if (i + 1 < n) { c1 = cand[i + 1]; l1 = lost[i + 1]; if (need_keys) k1 = items[i + 1]; }
    }
    t.key[2 * q] = k0; t.key[2 * q + 1] = k1;
    t.cnd[2 * q] = c0; t.cnd[2 * q + 1] = c1;
    t.flags |= ((c0 == 1u && l0 != tag) ? 1u : 0u) << (2 * q);
    t.flags |= ((c1 == 1u && l1 != tag) ? 1u : 0u) << (2 * q + 1);
  }
}

// rank of this thread's first item of every round INSIDE its segment + the segment totals into seg[] (one word per
// (round, wave), round-major): two ballots per round.  rank[q] is returned packed, 8 bits each (<= 128).
__device__ __forceinline__ uint32_t own_tile_rank(const OwnTile &t, uint32_t *seg) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t packed = 0;
#pragma unroll
  for (uint32_t q = 0; q < kOwnRounds; ++q) {
    const uint64_t m0 = __ballot((t.flags >> (2 * q)) & 1u), m1 = __ballot((t.flags >> (2 * q + 1)) & 1u);
    packed |= (uint32_t)(__popcll(m0 & lt) + __popcll(m1 & lt)) << (8 * q);
    if (lane == 0) seg[q * (kBlock / kWave) + wave] = (uint32_t)(__popcll(m0) + __popcll(m1));
  }
  return packed;
}

// owners park their keys in LDS at their rank inside the chunk, everybody writes its local ids (8-byte stores, a
// wave's store instruction covers one contiguous span)
template <bool BATCH>
__device__ __forceinline__ void own_tile_emit(const OwnTile &t, uint32_t prefix, uint32_t ranks_packed,
                                              const uint32_t *seg_excl /* exclusive prefix of the tile's segments */,
                                              uint32_t *s_keys, unsigned long long *__restrict__ w, uint32_t version,
                                              uint32_t *__restrict__ mapped, uint64_t tile0, uint64_t n, bool vec_ok) {
  const uint32_t wave = threadIdx.x >> 6;
#pragma unroll
  for (uint32_t q = 0; q < kOwnRounds; ++q) {
    uint32_t rank = seg_excl[q * (kBlock / kWave) + wave] + ((ranks_packed >> (8 * q)) & 0xffu); // inside the chunk
    uint32_t out[2];
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {
      const uint32_t e = 2 * q + j;
      out[j] = (BATCH && t.cnd[e] >= 2u) ? t.cnd[e] - 2u : kEmptyKey;
      if (t.flags & (1u << e)) {
        if (!BATCH) w[t.key[e]] = make_w1(version, 0u, prefix + rank);
        s_keys[rank] = t.key[e];
        out[j] = prefix + rank;
        ++rank;
      }
    }
    if (mapped) {
      const uint64_t i = tile0 + 2ull * kBlock * q + 2ull * threadIdx.x;
      if (vec_ok && i + 2 <= n) {
        *reinterpret_cast<uint2 *>(mapped + i) = make_uint2(out[0], out[1]);
      } else {
        if (i < n) mapped[i] = out[0];
        if (i + 1 < n) mapped[i + 1] = out[1];
      }
    }
  }
}

template <bool BATCH>
__global__ __launch_bounds__(kBlock) void k_owner_scan_chunked(unsigned long long *__restrict__ w, uint32_t version,
                                                               uint32_t *__restrict__ n2o,
                                                               const uint32_t *__restrict__ items,
                                                               const uint32_t *__restrict__ cand,
                                                               const unsigned long long *__restrict__ lost,
                                                               unsigned long long tag, uint32_t *__restrict__ mapped,
                                                               Count n_arg, unsigned long long *desc0,
                                                               uint32_t *chunk_desc, uint32_t epoch, uint32_t epoch16,
                                                               uint32_t *num_items, uint64_t *mirror_a,
                                                               uint64_t *mirror_b, uint32_t *err, uint32_t skip_below,
                                                               uint32_t patience, uint32_t delay0) {
  constexpr uint32_t FLAG_P = 2;
  __shared__ uint32_t smem[kBlock / kWave];
  __shared__ uint32_t s_total;
  __shared__ uint32_t s_miss[kChunkGrid / 32]; // chunks this workgroup counts itself (their word did not show up)
  __shared__ uint32_t s_seg[kChunkRegTiles * kOwnSegs];  // owners per (tile, round, wave) segment, then their prefix
  __shared__ uint32_t s_keys[kChunkRegTiles * kOwnTile]; // the owners' keys of a chunk (or of one tile), in rank order
  const uint64_t n = n_arg.get();
  const uint64_t tiles = (n + kOwnTile - 1) / kOwnTile;
  const uint32_t per = (uint32_t)((tiles + gridDim.x - 1) / gridDim.x);          // tiles per chunk
  const uint32_t chunks = per ? (uint32_t)((tiles + per - 1) / per) : 0u;        // chunks that hold items
  // 8-byte accesses of cand / items / mapped, 16-byte accesses of lost (workspace pieces are 16-byte aligned; a
  // caller's sliced tensor may not be)
  const bool vec_ok = (((((uintptr_t)items) | ((uintptr_t)cand) | ((uintptr_t)mapped)) & 7u) | (((uintptr_t)lost) & 15u)) == 0;
  // chunk = workgroup id, no ticket: 1024 tickets from one counter are 11 us of same-address atomics
  // (tools/micro_ticket.hip).  A chunk only ever waits for LOWER chunks, which publish before they wait for anybody;
  // nothing is assumed about when their workgroups run -- a chunk that has waited `patience` polls counts the missing
  // ones itself (below), so every resident workgroup finishes whatever else holds the device's slots.
  const uint32_t c = blockIdx.x;
  // ggms_debug_delay_next_scan (tests): chunk 0's workgroup "starts late" -- after the others have counted its chunk
  // themselves and the last one has replaced *num_items by the total
  if (c == 0 && delay0)
    for (uint32_t i = 0; i < delay0; ++i) __builtin_amdgcn_s_sleep(127);
  uint32_t base_seen = 0;
  unsigned long long d0_seen = 0;
  if (c == 0) { // the table's item count, and THEN whether somebody has computed this chunk's word already (see below)
    base_seen = __hip_atomic_load(num_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    d0_seen = __hip_atomic_load(desc0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (c < chunks && c >= skip_below) { // skip_below: 0; ggms_debug_poison_next_scan drops chunk 0 (tests of the bound)
    const uint64_t t0 = (uint64_t)c * per;
    const uint32_t my_tiles = (uint32_t)min((uint64_t)per, tiles - t0);
    const bool in_regs = per <= kChunkRegTiles; // uniform over the grid
    OwnTile tl[kChunkRegTiles];
    uint32_t ranks[kChunkRegTiles];
    uint32_t running = 0;
    if (in_regs) {
#pragma unroll
      for (uint32_t k = 0; k < kChunkRegTiles; ++k) {
        tl[k].flags = 0;
        if (k < my_tiles) own_tile_load(tl[k], items, cand, lost, tag, (t0 + k) * kOwnTile, n, vec_ok, true);
      }
#pragma unroll
      for (uint32_t k = 0; k < kChunkRegTiles; ++k) ranks[k] = own_tile_rank(tl[k], s_seg + k * kOwnSegs);
      __syncthreads();
      // exclusive prefix over the chunk's segments (tile-major, round, wave = item order), by wave 0
      if (threadIdx.x < kWave) {
        const uint32_t v = threadIdx.x < kChunkRegTiles * kOwnSegs ? s_seg[threadIdx.x] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (threadIdx.x < kChunkRegTiles * kOwnSegs) s_seg[threadIdx.x] = incl - v;
        if (threadIdx.x == 63) s_total = incl;
      }
      __syncthreads();
      running = s_total;
    } else { // count pass; the tiles are read again for the write-out
      uint32_t mine = 0;
      for (uint32_t k = 0; k < my_tiles; ++k) {
        OwnTile t;
        own_tile_load(t, items, cand, lost, tag, (t0 + k) * kOwnTile, n, vec_ok, false);
        mine += __popc(t.flags);
      }
      (void)block_exclusive_scan(mine, smem, running);
    }
    // publish this chunk's count; chunk 0 carries the table's item count in
    uint32_t prefix = 0;
    if (c == 0) {
      // base_seen was read before d0_seen: a chunk-0 word that is still missing then means the total (which needs that
      // word) had not replaced *num_items when it was read; a word that is there was computed by somebody who could
      // not wait for this workgroup -- from the same count, so the base is what it holds minus this chunk's owners
      prefix = base_seen;
      if ((uint32_t)(d0_seen >> 34) == epoch && ((uint32_t)(d0_seen >> 32) & 3u) == FLAG_P) prefix = (uint32_t)d0_seen - running;
      if (threadIdx.x == 0)
        __hip_atomic_store(desc0, scan_desc(epoch, FLAG_P, prefix + running), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (threadIdx.x == 0)
        __hip_atomic_store(&chunk_desc[c], (epoch16 << 16) | running, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // every earlier chunk's count: first all loads in flight together, then wait for the stragglers -- `patience`
      // polls; after that the workgroup counts the chunks that have not shown up itself (nothing is promised about
      // when their workgroups run, and this one holds a slot they may need)
      constexpr uint32_t kLook = kChunkGrid / kBlock;
      uint32_t got[kLook];
      bool ready[kLook];
#pragma unroll
      for (uint32_t r = 0; r < kLook; ++r) {
        const uint32_t j = threadIdx.x + r * kBlock;
        got[r] = 0;
        ready[r] = j == 0 || j >= c;
        if (j != 0 && j < c) {
          const uint32_t d = __hip_atomic_load(&chunk_desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ready[r] = (d >> 16) == epoch16;
          got[r] = d & 0xffffu;
        }
      }
      uint32_t got0 = 0; // thread 0: chunk 0's 64-bit word = absolute count (the base included)
      bool ready0 = threadIdx.x != 0;
      for (uint32_t spins = 0;; ++spins) {
        if (!ready0) {
          const unsigned long long d = __hip_atomic_load(desc0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(d >> 34) == epoch && ((uint32_t)(d >> 32) & 3u) == FLAG_P) { got0 = (uint32_t)d; ready0 = true; }
        }
        bool all = ready0;
#pragma unroll
        for (uint32_t r = 0; r < kLook; ++r) all &= ready[r];
        if (__syncthreads_and(all)) break;
        if (spins >= patience) { // uniform: serve ourselves
          if (threadIdx.x < kChunkGrid / 32) s_miss[threadIdx.x] = 0;
          __syncthreads();
#pragma unroll
          for (uint32_t r = 0; r < kLook; ++r) {
            const uint32_t j = threadIdx.x + r * kBlock;
            if (!ready[r]) atomicOr(&s_miss[j >> 5], 1u << (j & 31u));
          }
          if (!ready0) atomicOr(&s_miss[0], 1u);
          __syncthreads();
          for (uint32_t wd = 0; wd < kChunkGrid / 32; ++wd) {
            uint32_t m = s_miss[wd];
            while (m) {
              const uint32_t j = wd * 32 + (uint32_t)__builtin_ctz(m);
              m &= m - 1u;
              // chunk j's owners, counted by this workgroup from the scan's input (nobody rewrites cand / lost)
              const uint64_t tj = (uint64_t)j * per;
              const uint32_t ntj = (uint32_t)min((uint64_t)per, tiles - tj);
              uint32_t mine = 0, cnt_j;
              for (uint32_t k = 0; k < ntj; ++k) {
                OwnTile t;
                own_tile_load(t, items, cand, lost, tag, (tj + k) * kOwnTile, n, vec_ok, false);
                mine += __popc(t.flags);
              }
              (void)block_exclusive_scan(mine, smem, cnt_j);
              if (j == 0) {
                if (threadIdx.x == 0) {
                  const uint32_t v1 = __hip_atomic_load(num_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // the count was READ before the word is looked at again
                  const unsigned long long d = __hip_atomic_load(desc0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  if ((uint32_t)(d >> 34) == epoch && ((uint32_t)(d >> 32) & 3u) == FLAG_P) {
                    got0 = (uint32_t)d;
                  } else { // still missing: the total (which needs this word) had not replaced *num_items at v1's time
                    got0 = v1 + cnt_j;
                    __hip_atomic_store(desc0, scan_desc(epoch, FLAG_P, got0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // out before this chunk (if it is the last) writes the total
                  }
                  ready0 = true;
                }
              } else {
#pragma unroll
                for (uint32_t r = 0; r < kLook; ++r)
                  if (threadIdx.x + r * kBlock == j) {
                    got[r] = cnt_j;
                    ready[r] = true;
                    __hip_atomic_store(&chunk_desc[j], (epoch16 << 16) | cnt_j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  }
              }
            }
          }
          break;
        }
        if (spins > (1u << 22)) { // no help possible (patience switched off) and a predecessor never showed up
          if (!all && err) atomicOr(err, kErrScanSpin);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (uint32_t r = 0; r < kLook; ++r)
          if (!ready[r]) {
            const uint32_t d = __hip_atomic_load(&chunk_desc[threadIdx.x + r * kBlock], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ready[r] = (d >> 16) == epoch16;
            got[r] = ready[r] ? (d & 0xffffu) : 0u;
          }
      }
      uint32_t sum = got0;
#pragma unroll
      for (uint32_t r = 0; r < kLook; ++r) sum += got[r];
      (void)block_exclusive_scan(sum, smem, prefix);
    }
    if (c + 1 == chunks && threadIdx.x == 0) { // the last chunk knows the total
      const uint32_t total = prefix + running;
      *num_items = total;
      if (mirror_a) *mirror_a = (uint64_t)total;
      if (mirror_b) *mirror_b = (uint64_t)total;
    }
    if (in_regs) {
#pragma unroll
      for (uint32_t k = 0; k < kChunkRegTiles; ++k)
        if (k < my_tiles)
          own_tile_emit<BATCH>(tl[k], prefix, ranks[k], s_seg + k * kOwnSegs, s_keys, w, version, mapped,
                               (t0 + k) * kOwnTile, n, vec_ok);
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < running; i += kBlock) n2o[prefix + i] = s_keys[i]; // the chunk's slice of n2o
    } else {
      uint32_t run2 = prefix;
      for (uint32_t k = 0; k < my_tiles; ++k) {
        OwnTile t;
        own_tile_load(t, items, cand, lost, tag, (t0 + k) * kOwnTile, n, vec_ok, true);
        const uint32_t rk = own_tile_rank(t, s_seg);
        __syncthreads();
        if (threadIdx.x < kWave) {
          const uint32_t v = threadIdx.x < kOwnSegs ? s_seg[threadIdx.x] : 0u;
          const uint32_t incl = wave_inclusive_scan(v);
          if (threadIdx.x < kOwnSegs) s_seg[threadIdx.x] = incl - v;
          if (threadIdx.x == 63) s_total = incl;
        }
        __syncthreads();
        const uint32_t total = s_total;
        own_tile_emit<BATCH>(t, run2, rk, s_seg, s_keys, w, version, mapped, (t0 + k) * kOwnTile, n, vec_ok);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < total; i += kBlock) n2o[run2 + i] = s_keys[i];
        __syncthreads(); // s_seg / s_keys / s_total are rewritten by the next tile
        run2 += total;
      }
    }
  }
  if (threadIdx.x == 0 && tiles == 0 && c == 0) { // empty input: the count stays what it is
    const uint32_t base = *num_items;
    if (mirror_a) *mirror_a = (uint64_t)base;
    if (mirror_b) *mirror_b = (uint64_t)base;
  }
}

static std::atomic<bool> g_poison_next_scan{false};

constexpr size_t kOwnerScanTileGrid = 512; // workgroups of the tile-chained form

// descriptors the owner scan needs for n_max items (64-bit words behind the 8 control words)
size_t owner_scan_tiles(size_t n_max) { return (n_max + kOwnTile - 1) / kOwnTile; }

// workspace of one fill: cand / item_pos [n], lost [n] (64-bit), scan area
size_t ht_ws_words(size_t num_input) { return 3 * num_input + tile_scan_words(num_input) + 24 + chunk_desc_words(); }
size_t chunk_desc_words() { return kChunkGrid + 8; }

__global__ void k_status_copy(uint32_t *status, uint32_t *device_word, uint64_t *status_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) take_batch_status(status, device_word, status_out);
}

// status_out != NULL: this is the batch's last launch -- it takes the batch's status word (ht->num_items_dev + 1)
int launch_map_rest_all(const ggms_hashtable_t *ht, const MapRestJobs &jobs, uint32_t num_jobs, size_t max_items,
                        const IdxMap &map, uint64_t *status_out, hipStream_t s) {
  uint32_t *batch_status = status_out ? ht->num_items_dev + 1 : nullptr;
  uint32_t *device_word = status_out ? device_status_word() : nullptr;
  if (num_jobs == 0) { // nothing deferred (hashed layout, or no layer could sample): only the status word
    if (!status_out) return GGMS_OK;
    hipLaunchKernelGGL(k_status_copy, dim3(1), dim3(64), 0, s, batch_status, device_word, status_out);
    GGMS_LAUNCH_CHECK();
    return GGMS_OK;
  }
  const int gx = grid_for(max_items ? max_items : 1, kBlock);
  hipLaunchKernelGGL(k_map_rest_all, dim3(gx, num_jobs), dim3(kBlock), 0, s, (const unsigned long long *)ht->o2n, jobs, map,
                     batch_status, device_word, status_out);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

int seed_enter_impl(const ggms_hashtable_t *ht, const uint32_t *seeds, size_t num_seeds, const BatchPrologue &pro,
                    hipStream_t s) {
  hipLaunchKernelGGL(k_seed_enter, dim3(grid_for(num_seeds, kBlock)), dim3(kBlock), 0, s, (unsigned long long *)ht->o2n,
                     ht->version, seeds, count_of(num_seeds), ht->n2o, pro);
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
  BatchPrologue pro{nullptr, 0, nullptr, 0, nullptr, nullptr};
  if (prologue) pro = *prologue;
  if (n_max == 0) { // no kernel to ride on
    if (pro.num_zero) GGMS_HIP(hipMemsetAsync(pro.zero_words, 0, pro.num_zero * sizeof(uint32_t), s));
    if (pro.num_zero2) GGMS_HIP(hipMemsetAsync(pro.zero_words2, 0, pro.num_zero2 * sizeof(uint32_t), s));
    if (pro.num_items) GGMS_HIP(hipMemsetAsync(pro.num_items, 0, sizeof(uint32_t), s));
    if (pro.record_n) GGMS_HIP(hipMemsetAsync(pro.record_n, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  Table t = table_of(ht);
  uint32_t *err = scratch.status_word(); // the batch's own word, or the device's sticky one (leaf operators)
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
  }
  // ggms_debug_poison_next_scan (tests): tile / chunk 0 is never processed, so every later one waits for it until its
  // bound -- the failure the status word exists for
  const bool poisoned = !scratch.cleared && g_poison_next_scan.exchange(false);
  if (poisoned) GGMS_HIP(hipMemsetD32Async((hipDeviceptr_t)ctl, 1, 1, s));
  // chunked form (no ticket, one look-back per workgroup) wherever the caller's scan area has room for its
  // descriptors.  Test aids (ggms_debug_set_knob): the tile-chained kernel, which otherwise only runs beyond 65 M
  // items, and fewer chunks than kChunkGrid (chunks too long for registers take the two-pass form)
  const bool force_tiles = debug_knob(GGMS_DEBUG_OWNER_SCAN_TILES) > 0;
  const long long knob_chunks = debug_knob(GGMS_DEBUG_OWNER_SCAN_CHUNKS);
  const size_t chunk_grid = knob_chunks > 0 && knob_chunks < (long long)kChunkGrid ? (size_t)knob_chunks : (size_t)kChunkGrid;
  const int cgrid = (int)std::min<size_t>(std::max<size_t>(owner_scan_tiles(n_max), 1), chunk_grid);
  // a chunk's owner count travels in 16 bits: at most kChunkMaxTiles tiles per chunk (65 M items at the full grid);
  // beyond that the tile-chained kernel runs
  const bool chunk_fits = owner_scan_tiles(n_max) <= (size_t)kChunkMaxTiles * (size_t)cgrid;
  if (scratch.chunk && !force_tiles && chunk_fits) {
    if (!scratch.cleared) GGMS_HIP(hipMemsetAsync(scratch.chunk, 0, chunk_desc_words() * sizeof(uint32_t), s));
    const uint32_t epoch = next_scan_epoch();
    const uint32_t epoch16 = epoch % 65535u + 1u; // never 0: a cleared descriptor is "not written"
    const uint32_t skip_below = poisoned ? 1u : 0u;
    const uint32_t patience = poisoned ? kNoPatienceLimit : scan_patience(); // the poisoned launch tests the BOUND
    const uint32_t delay0 = scan_delay_word().exchange(0u);
    unsigned long long *desc0 = desc;             // chunk 0: 64-bit, carries the table's item count in
    if (di.batch)
      hipLaunchKernelGGL(k_owner_scan_chunked<true>, dim3(cgrid), dim3(kBlock), 0, s, di.w, di.version, ht->n2o, input,
                         di.cand, di.lost, di.tag, mapped, n, desc0, scratch.chunk, epoch, epoch16, ht->num_items_dev,
                         mirror_a, mirror_b, err, skip_below, patience, delay0);
    else
      hipLaunchKernelGGL(k_owner_scan_chunked<false>, dim3(cgrid), dim3(kBlock), 0, s, di.w, di.version, ht->n2o, input,
                         di.cand, di.lost, di.tag, mapped, n, desc0, scratch.chunk, epoch, epoch16, ht->num_items_dev,
                         mirror_a, mirror_b, err, skip_below, patience, delay0);
  } else {
    // Fewer workgroups than tiles on purpose: a workgroup takes tiles from the ticket one after the other, so by the
    // time tile t is taken the tiles before t - grid have finished and the look-back finds a published prefix in
    // its first window; with one workgroup per tile every tile starts at once and tile t walks t / 64 windows.
    const int oscan_grid = (int)std::min<size_t>(grid_for(owner_scan_tiles(n_max), 1), kOwnerScanTileGrid);
    if (di.batch)
      hipLaunchKernelGGL(k_owner_scan<true>, dim3(oscan_grid), dim3(kBlock), 0, s, di.w, di.version, ht->n2o, input, di.cand,
                         di.lost, di.tag, mapped, n, ctl, desc, next_scan_epoch(), ht->num_items_dev, mirror_a, mirror_b, err);
    else
      hipLaunchKernelGGL(k_owner_scan<false>, dim3(oscan_grid), dim3(kBlock), 0, s, di.w, di.version, ht->n2o, input, di.cand,
                         di.lost, di.tag, mapped, n, ctl, desc, next_scan_epoch(), ht->num_items_dev, mirror_a, mirror_b, err);
  }
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
void ggms_debug_set_scan_patience(uint32_t polls) { scan_patience_word().store(polls); }
void ggms_debug_delay_next_scan(uint32_t sleeps) { scan_delay_word().store(sleeps); }

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
  GGMS_HIP(hipMemsetAsync(ht->num_items_dev, 0, 2 * sizeof(uint32_t), to_stream(stream))); // item count + batch status
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
    ScanArea area{scan_words, false};
    area.chunk = scan_words + tile_scan_words(num_input) + 8; // the last chunk_desc_words() of the workspace
    DedupInsert di{};
    di.cand = item_pos;
    di.lost = lost; // leaf mode: base 0, the owners' words are rewritten as {assigned, local id}
    int rc = ht_fill_impl(ht, input, num_input, count_of(num_input), di, false, area, nullptr, nullptr, s, nullptr, nullptr,
                          kRestNow, nullptr);
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
