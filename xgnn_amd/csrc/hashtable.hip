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
#include "ggms_internal.h"
#include "tile_scan.h"

namespace ggms {

__device__ __forceinline__ unsigned long long make_w0(uint32_t version, uint32_t key) {
  return ((unsigned long long)key << 32) | version;
}
__device__ __forceinline__ uint32_t w0_version(unsigned long long w) { return (uint32_t)w; }
__device__ __forceinline__ unsigned long long make_w1(uint32_t version, uint32_t pending, uint32_t value) {
  const unsigned long long hi = ((unsigned long long)(0x7fffffffu - version) << 1) | pending;
  return (hi << 32) | value;
}

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

  // hashed only: bucket position of `key`, inserting it if absent
  __device__ __forceinline__ uint32_t find_or_claim(uint32_t key) const {
    uint32_t pos = key & mask;
    uint32_t delta = 1;
    const unsigned long long want = make_w0(version, key);
    for (;;) {
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
                                                      uint32_t *__restrict__ item_pos, BatchPrologue pro) {
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
    uint32_t pos = key;
    if (!DIRECT) {
      pos = t.find_or_claim(key);
      item_pos[i] = pos;
    }
    atomicMin(t.w1<DIRECT>(pos), make_w1(t.version, 1u, (uint32_t)i));
  }
}

// count_hashmap / compact_hashmap (cuda_hashtable.cu:197-232, 406-458):
// instance i owns its key iff the word still says {pending, i}
// hint (optional, direct layout): hint[i] == 0 means the inserting atomicMin already saw a smaller word, i.e. an earlier
// instance or an assigned id -- instance i cannot own the key and its word need not be read again
template <bool DIRECT>
struct OwnerFlag {
  Table t;
  const uint32_t *item_pos; // hashed: bucket positions; direct: the keys themselves
  const uint32_t *hint;
  __device__ __forceinline__ uint32_t operator()(uint64_t i) const {
    if (DIRECT && hint && hint[i] == 0u) return 0u;
    return *t.w1<DIRECT>(item_pos[i]) == make_w1(t.version, 1u, (uint32_t)i) ? 1u : 0u;
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

// the instances that do not own their key (AssignLocal left kEmptyKey): read the owner's local id
template <bool DIRECT>
__global__ __launch_bounds__(kBlock) void k_map_rest(Table t, const uint32_t *__restrict__ item_pos, Count n_arg,
                                                     uint32_t *__restrict__ out) {
  const uint64_t n = n_arg.get();
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    if (out[i] == kEmptyKey) out[i] = (uint32_t)*t.w1<DIRECT>(item_pos[i]);
}

size_t ht_ws_words(size_t num_input) { return num_input + tile_scan_words(num_input) + 16; }

// mapped != NULL: also produce the local id of every input instance (FillWithDuplicates + the dst half of
// GPUMapEdges in one go): owners write theirs while assigning, k_map_rest looks up the rest.
// clear_area: this is the first kernel of a batch -- it also zeroes the shared scan area's control words.
int ht_fill_impl(const ggms_hashtable_t *ht, const uint32_t *input, size_t n_max, Count n, uint32_t *item_pos,
                 ScanArea scratch, uint64_t *mirror_a, uint64_t *mirror_b, hipStream_t s, uint32_t *mapped,
                 const BatchPrologue *prologue, bool inserted) {
  BatchPrologue pro{nullptr, 0, nullptr, nullptr};
  if (prologue) pro = *prologue;
  if (n_max == 0) { // no kernel to ride on
    if (pro.num_zero) GGMS_HIP(hipMemsetAsync(pro.zero_words, 0, pro.num_zero * sizeof(uint32_t), s));
    if (pro.num_items) GGMS_HIP(hipMemsetAsync(pro.num_items, 0, sizeof(uint32_t), s));
    if (pro.record_n) GGMS_HIP(hipMemsetAsync(pro.record_n, 0, sizeof(uint64_t), s));
    return GGMS_OK;
  }
  Table t = table_of(ht);
  const int grid = grid_for(n_max, kBlock);
  int rc;
  if (ht->direct) {
    if (!inserted) {
      hipLaunchKernelGGL(k_ht_insert<true>, dim3(grid), dim3(kBlock), 0, s, t, input, n, item_pos, pro);
      GGMS_LAUNCH_CHECK();
    }
    scratch.stash = item_pos; // the direct layout does not use item_pos: it holds the owner flags between passes
    rc = tile_scan(OwnerFlag<true>{t, input, inserted ? item_pos : nullptr}, AssignLocal<true>{t, input, input, mapped},
                   n_max, n, scratch,
                   ht->num_items_dev, ht->num_items_dev, nullptr, s, mirror_a, mirror_b);
  } else {
    hipLaunchKernelGGL(k_ht_insert<false>, dim3(grid), dim3(kBlock), 0, s, t, input, n, item_pos, pro);
    GGMS_LAUNCH_CHECK();
    rc = tile_scan(OwnerFlag<false>{t, item_pos, nullptr}, AssignLocal<false>{t, input, item_pos, mapped}, n_max, n, scratch,
                   ht->num_items_dev, ht->num_items_dev, nullptr, s, mirror_a, mirror_b);
  }
  if (rc != GGMS_OK || !mapped) return rc;
  if (ht->direct)
    hipLaunchKernelGGL(k_map_rest<true>, dim3(grid), dim3(kBlock), 0, s, t, input, n, mapped);
  else
    hipLaunchKernelGGL(k_map_rest<false>, dim3(grid), dim3(kBlock), 0, s, t, item_pos, n, mapped);
  GGMS_LAUNCH_CHECK();
  return GGMS_OK;
}

} // namespace ggms

using namespace ggms;

extern "C" {

// TableSize(num, scale = kDefaultScale = 2): cuda_hashtable.cu:146-149, cuda_hashtable.h:105
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
  // item_pos[num_input] + tile scan scratch
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
    int rc = ht_fill_impl(ht, input, num_input, count_of(num_input), item_pos,
                          ScanArea{item_pos + num_input, false}, nullptr, nullptr, s, nullptr, nullptr);
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
