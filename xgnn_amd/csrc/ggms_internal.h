// ggms_internal.h -- cross-file C++ entry points (count-on-device variants of the leaf ops).
#pragma once
#include "tile_scan.h"

namespace ggms {

// What goes into out_src: the seed's global id (leaf API, like the reference) or
// its local id (fused batch path: the COO `col` is written by the sampler itself).
struct SrcMode {
  const uint32_t *seed_local; // local id per seed position; NULL = the position itself
  int local;
  __device__ __forceinline__ uint32_t value(uint32_t rid, uint64_t index) const {
    if (!local) return rid;
    return seed_local ? seed_local[index] : (uint32_t)index;
  }
};

// What the first kernel of a batch (the insert of the seeds) does on the side: clear the shared scan area's
// control words + single-pass descriptors, reset the table's item count, record |seeds|.
struct BatchPrologue {
  uint32_t *zero_words;
  uint32_t num_zero;
  uint32_t *zero_words2; // second range (the chunked owner scan's descriptors)
  uint32_t num_zero2;
  uint32_t *num_items;
  uint64_t *record_n;
  uint32_t *zero_words3 = nullptr; // third range (fused first layer: the descriptors behind the ones it uses itself)
  uint32_t num_zero3 = 0;
  uint32_t items_are_seeds = 0;    // distinct seeds: the table's item count starts at |seeds| instead of 0
  __device__ __forceinline__ void run(uint64_t n, uint32_t tid, uint32_t nthreads) const { // by ONE workgroup
    for (uint32_t z = tid; z < num_zero; z += nthreads) zero_words[z] = 0u;
    for (uint32_t z = tid; z < num_zero2; z += nthreads) zero_words2[z] = 0u;
    for (uint32_t z = tid; z < num_zero3; z += nthreads) zero_words3[z] = 0u;
    if (tid == 0) {
      if (num_items) *num_items = items_are_seeds ? (uint32_t)n : 0u;
      if (record_n) *record_n = n;
    }
  }
};
// distinct seeds entered by the first layer's own launch (khop3): the batch prologue rides on it, the seeds become
// the head of the unique list where they are read
struct FirstLayer {
  BatchPrologue pro;
  uint32_t *n2o;
};
// sample_khop.hip
size_t sample_ws_words(size_t num_input);
size_t khop0_ws_words(size_t num_input, size_t fanout);
int sample_khop3_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t fanout, uint32_t *out_src,
                      uint32_t *out_dst, uint64_t *num_out_dev, uint32_t *states, uint32_t *workspace,
                      const uint32_t *seed_local, int src_local, hipStream_t s, ScanArea *shared_scan = nullptr,
                      const DedupInsert *insert = nullptr, const FirstLayer *first = nullptr);
bool khop3_can_fuse_seeds(size_t num_seeds); // every tile of the first layer has a workgroup of its own
// distinct seeds, any sampler: one launch -- table entries, head of the unique list, batch prologue (hashtable.hip)
int seed_enter_impl(const ggms_hashtable_t *ht, const uint32_t *seeds, size_t num_seeds, const BatchPrologue &pro,
                    hipStream_t s);
// distinct seeds entered by khop0's plan pass (the first kernel of the batch): table words, head of the unique list,
// batch prologue -- k_seed_enter's work without its launch
struct SeedEnter {
  unsigned long long *w;
  uint32_t version;
  uint32_t *n2o;
  BatchPrologue pro;
};
bool khop0_can_enter_seeds(size_t num_seeds); // its plan pass is one launch
size_t khop0_plan_desc_words(size_t num_seeds); // descriptor words of the batch's scan area that launch uses itself
int sample_khop0_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t fanout, uint32_t *out_src,
                      uint32_t *out_dst, uint64_t *num_out_dev, uint32_t *workspace, const uint32_t *seed_local,
                      int src_local, hipStream_t s, ScanArea *shared_scan = nullptr, const DedupInsert *insert = nullptr,
                      const SeedEnter *enter = nullptr);
int sample_khop2_impl(const uint32_t *indptr, uint32_t *indices, size_t num_node, const uint32_t *input, size_t n_max,
                      Count n, uint32_t fanout, uint32_t *out_src, uint32_t *out_dst, uint64_t *num_out_dev,
                      uint32_t *states, uint32_t *workspace, const uint32_t *seed_local, int src_local, hipStream_t s,
                      ScanArea *shared_scan = nullptr);

// sample_weighted.hip
size_t weighted_ws_words(size_t num_input, size_t fanout);
int sample_weighted_impl(const uint32_t *indptr, const uint32_t *indices, const float *prob, const uint32_t *alias,
                         const uint32_t *input, size_t n_max, Count n, uint32_t fanout, uint32_t *out_src,
                         uint32_t *out_dst, uint64_t *num_out_dev, uint32_t *states, uint32_t *workspace,
                         const uint32_t *seed_local, int src_local, hipStream_t s, ScanArea *shared_scan = nullptr,
                         uint32_t num_node = 0, const DedupInsert *insert = nullptr);
// sample_random_walk.hip
size_t random_walk_ws_words(size_t num_input, size_t walk_length, size_t num_walk, size_t K);
int sample_random_walk_impl(GraphView g, const uint32_t *input, size_t n_max, Count n, uint32_t walk_length,
                            double restart_prob, uint32_t num_walk, uint32_t K, uint32_t *out_src, uint32_t *out_dst,
                            uint32_t *out_data, uint64_t *num_out_dev, uint32_t *states, uint32_t *workspace,
                            const uint32_t *seed_local, int src_local, hipStream_t s, ScanArea *shared_scan,
                            const DedupInsert *insert);
size_t walk_scan_tiles(size_t num_input);

int sample_weighted_hash_dedup_impl(const uint32_t *indptr, const uint32_t *indices, const float *prob,
                                    const uint32_t *alias, const uint32_t *input, size_t n_max, Count n,
                                    uint32_t fanout, uint32_t *out_src, uint32_t *out_dst, uint64_t *num_out_dev,
                                    uint32_t *states, uint32_t *workspace, const uint32_t *seed_local, int src_local,
                                    hipStream_t s, ScanArea *shared_scan = nullptr, const DedupInsert *insert = nullptr);

// hashtable.hip
size_t ht_ws_words(size_t num_input);
size_t chunk_desc_words(); // 32-bit descriptors of the chunked owner scan: their own piece of a scan area
// the end-of-batch id look-ups of the instances that do not own their key: one job per layer
struct MapRestJobs {
  uint32_t *row[16];
  const uint32_t *key[16];
  const uint64_t *num[16];
};
int launch_map_rest_all(const ggms_hashtable_t *ht, const MapRestJobs &jobs, uint32_t num_jobs, size_t max_items,
                        const IdxMap &map, uint64_t *status_out, hipStream_t s);
size_t owner_scan_tiles(size_t n_max);
// insert + ordered local-id assignment (see DedupInsert in ggms_device.h for the two modes).
// di: cand (n_max words; hashed layout: bucket positions) + lost (n_max 64-bit words), and in batch mode the index
//     base of this fill + the IdxMap of the fills before it; w / version / tag are filled in here.
// inserted: the producer of `input` already entered every item with exactly this `di` (fused sampler).
// mirror_a/b (optional): 64-bit device slots that also receive the new item count.
// mapped (optional): mapped[i] = local id of input[i] (the dst half of GPUMapEdges, fused).
// rest: kRestNow resolves the instances that do not own their key here (batch mode: needs n_dev_for_rest, a device
//       copy of the item count, and di.map must already include this fill); kRestDefer leaves kEmptyKey for the
//       caller's launch_map_rest_all at the end of the batch.
constexpr int kRestNow = 0, kRestDefer = 1;
int ht_fill_impl(const ggms_hashtable_t *ht, const uint32_t *input, size_t n_max, Count n, DedupInsert di,
                 bool inserted, ScanArea scan, uint64_t *mirror_a, uint64_t *mirror_b, hipStream_t s, uint32_t *mapped,
                 const BatchPrologue *prologue, int rest, const uint64_t *n_dev_for_rest);

} // namespace ggms
