// sample_batch.hip -- one mini-batch of k-layer sampling, enqueued in one go.
//
// Reference: DoGPUSample (dist/dist_loops.cc:62-368; single-process twin
// cuda/cuda_loops.cc:54-292): per layer {sample -> D2H nnz -> FillWithDuplicates ->
// D2H num_unique -> GPUMapEdges}, i.e. two host round trips per layer plus one
// StreamSync after every kernel.
//
// Here the whole loop is a straight sequence of launches on one stream.  Every
// size that the next launch depends on (frontier size, edge count) stays in
// device memory and is read by the kernels themselves (ggms::Count); grids are
// sized from the closed-form upper bounds of PredictNumNodes (common.cc:488-497)
// and grid-stride, so a launch never needs the exact value.  The caller syncs
// once, after the feature extract that follows.
//
// Other savings against the reference, all result-preserving:
//   * the frontier of layer i-1 is the hash table's n2o prefix: no `unique`
//     buffer is allocated or copied (dist_loops.cc:271-280,350-353);
//   * `col` (local id of the edge's seed) is written by the sampler: a seed's
//     local id is its position in the frontier (n2o), so the E hash lookups of
//     GPUMapEdges' src half (cuda_mapping.cu:57-60) reduce to none (n lookups
//     for the first layer, whose input is the raw seed list);
//   * `row` (local id of the sampled neighbour) is produced by the table fill itself:
//     the instance that owns a key gets its id in the owner scan, an instance that
//     loses to an EARLIER layer resolves its id inside the sampler, and the few that
//     lose inside their own layer are looked up for all layers at once at the end
//     (k_map_rest_all) -- the dst half of GPUMapEdges without a pass of random reads
//     over all E edges, and without rewriting the table (batch mode, ggms_device.h);
//   * with the direct table layout khop3's fused launch also enters the neighbours
//     into the table (one returning atomicMin each), so FillWithDuplicates is only
//     the owner scan: 3 + 2 L + 1 launches per batch in all;
//   * the batch prologue (scan-area clear, item-count reset, |seeds| record) rides on
//     the first kernel of the batch instead of three tiny launches.
// Several batches may be in flight on different streams (own table + workspace each):
// ggms_sample_extra_t.rng_wait / rng_done chain only the sampler kernels, which share
// the RNG pool, in batch order.
#include <algorithm>

#include "ggms_internal.h"
#include "tile_scan.h"

namespace ggms {

__global__ void k_record(uint64_t *slot, Count c) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *slot = c.get();
}

struct BatchCaps {
  size_t max_input[16];
  size_t max_edges[16];
  size_t max_in_all, max_e_all;
};

static BatchCaps caps_of(size_t num_seeds, const size_t *fanouts, uint32_t L) {
  BatchCaps c{};
  size_t n = num_seeds;
  for (int i = (int)L - 1; i >= 0; --i) {
    c.max_input[i] = n;
    c.max_edges[i] = n * fanouts[i];
    n += c.max_edges[i]; // unique nodes so far <= previous unique + new edges
    if (c.max_input[i] > c.max_in_all) c.max_in_all = c.max_input[i];
    if (c.max_edges[i] > c.max_e_all) c.max_e_all = c.max_edges[i];
  }
  return c;
}

} // namespace ggms

using namespace ggms;

extern "C" {

int ggms_event_create(ggms_event_t *event) {
  GGMS_CHECK_ARG(event);
  hipEvent_t e;
  GGMS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = (ggms_event_t)e;
  return GGMS_OK;
}

int ggms_event_destroy(ggms_event_t event) {
  if (event) GGMS_HIP(hipEventDestroy((hipEvent_t)event));
  return GGMS_OK;
}

int ggms_sample_batch_capacity(size_t num_seeds, const size_t *fanouts, uint32_t num_layer, size_t *max_input,
                               size_t *max_edges, size_t *max_unique) {
  GGMS_CHECK_ARG(fanouts && num_layer >= 1 && num_layer <= 16);
  const BatchCaps c = caps_of(num_seeds, fanouts, num_layer);
  for (uint32_t i = 0; i < num_layer; ++i) {
    if (max_input) max_input[i] = c.max_input[i];
    if (max_edges) max_edges[i] = c.max_edges[i];
  }
  if (max_unique) *max_unique = c.max_input[0] + c.max_edges[0];
  return GGMS_OK;
}

static size_t sampler_ws_words(int sample_type, const BatchCaps &c, const size_t *fanouts, uint32_t L,
                               const ggms_sample_extra_t *extra) {
  size_t w = sample_ws_words(c.max_in_all);
  for (uint32_t i = 0; i < L; ++i) {
    if (sample_type == GGMS_KHOP0) w = std::max(w, khop0_ws_words(c.max_input[i], fanouts[i]));
    if (sample_type == GGMS_WEIGHTED_KHOP || sample_type == GGMS_KHOP1 || sample_type == GGMS_WEIGHTED_KHOP_PREFIX)
      w = std::max(w, weighted_ws_words(c.max_input[i], fanouts[i]));
    if (sample_type == GGMS_RANDOM_WALK && extra)
      w = std::max(w, random_walk_ws_words(c.max_input[i], extra->random_walk_length, extra->num_random_walk, fanouts[i]));
  }
  return w;
}

// ---- workspace layout of one batch (uint32 words; every piece starts 16-byte aligned) --------------------------
struct BatchLayout {
  size_t seed_local, samp_ws, tmp_dst[16], cand, lost, scan, chunk, total;
  size_t dedup_items; // entries of cand / lost
  size_t scan_tiles;  // descriptors the scans of the batch may use (cleared by the batch prologue)
};
static inline size_t up4(size_t w) { return (w + 3) & ~(size_t)3; }

static BatchLayout layout_of(int sample_type, size_t num_seeds, const size_t *fanouts, uint32_t L, const BatchCaps &c,
                             const ggms_sample_extra_t *extra) {
  BatchLayout l{};
  size_t w = 0;
  l.seed_local = w;  w += up4(num_seeds + 16);
  l.samp_ws = w;     w += up4(sampler_ws_words(sample_type, c, fanouts, L, extra));
  for (uint32_t i = 0; i < L; ++i) { // global neighbour ids of every layer: kept for the end-of-batch id look-ups
    l.tmp_dst[i] = w;
    w += up4(c.max_edges[i] + 16);
  }
  l.dedup_items = std::max(c.max_e_all, num_seeds);
  l.cand = w;        w += up4(l.dedup_items + 16);          // hashed layout: bucket positions
  l.lost = w;        w += up4(2 * (l.dedup_items + 16));    // 64-bit tags
  // ONE scan area for the batch: sampler offsets (tiles of 128 seeds), owner scans (tiles of 2048 items), the
  // generic tile scans of the other samplers (<= kSinglePassTiles descriptors, else three launches)
  l.scan_tiles = std::max<size_t>({c.max_in_all / 128 + 2, owner_scan_tiles(l.dedup_items) + 2, kSinglePassTiles + 2});
  if (sample_type == GGMS_RANDOM_WALK) l.scan_tiles = std::max(l.scan_tiles, walk_scan_tiles(c.max_in_all));
  if (sample_type == GGMS_KHOP0) l.scan_tiles = std::max<size_t>(l.scan_tiles, 2 * (kSinglePassTiles + 2)); // two sums per pass
  l.scan = w;
  w += up4(std::max(tile_scan_words(std::max(c.max_e_all, c.max_in_all)), 8 + 2 * l.scan_tiles + 4) + 16);
  // the chunked owner scan's 32-bit descriptors + one ticket set per ticketed sampler launch: a piece of their own,
  // cleared by the batch prologue too
  l.chunk = w;
  w += up4(chunk_desc_words() + (size_t)kTicketSets * kTicketWords);
  l.total = w;
  return l;
}

size_t ggms_sample_batch_workspace_bytes(int sample_type, size_t num_seeds, const size_t *fanouts, uint32_t num_layer,
                                         const ggms_sample_extra_t *extra) {
  if (!fanouts || num_layer < 1 || num_layer > 16) return 0;
  const BatchCaps c = caps_of(num_seeds, fanouts, num_layer);
  return layout_of(sample_type, num_seeds, fanouts, num_layer, c, extra).total * sizeof(uint32_t) + 16;
}

int ggms_sample_batch(int sample_type, const ggms_graph_t *graph, const ggms_id_t *seeds, size_t num_seeds,
                      const size_t *fanouts, uint32_t num_layer, ggms_hashtable_t *ht, void *states,
                      size_t num_states, ggms_id_t *const *row, ggms_id_t *const *col, uint64_t *counts_dev,
                      const ggms_sample_extra_t *extra, void *workspace, size_t workspace_bytes,
                      ggms_stream_t stream) {
  GGMS_CHECK_ARG(graph && fanouts && ht && row && col && counts_dev);
  GGMS_CHECK_ARG(num_layer >= 1 && num_layer <= 16);
  GGMS_CHECK_ARG(sample_type >= GGMS_KHOP0 && sample_type <= GGMS_KHOP3);
  GGMS_CHECK_ARG(num_seeds == 0 || seeds);
  for (uint32_t i = 0; i < num_layer; ++i) GGMS_CHECK_ARG(fanouts[i] > 0); // a layer that samples nothing is a config error
  GGMS_CHECK_ARG(workspace && workspace_bytes >= ggms_sample_batch_workspace_bytes(sample_type, num_seeds, fanouts,
                                                                                   num_layer, extra));
  if (sample_type == GGMS_WEIGHTED_KHOP)
    GGMS_CHECK_ARG(extra && extra->prob_table && extra->alias_table && graph->num_part == 0 && states);
  if (sample_type == GGMS_KHOP1) GGMS_CHECK_ARG(graph->num_part == 0 && graph->indptr && graph->indices && states);
  if (sample_type == GGMS_WEIGHTED_KHOP_PREFIX)
    GGMS_CHECK_ARG(extra && extra->prob_table && graph->num_part == 0 && states);
  if (sample_type == GGMS_WEIGHTED_KHOP_HASH_DEDUP) {
    GGMS_CHECK_ARG(extra && extra->prob_table && extra->alias_table && graph->num_part == 0);
    GGMS_CHECK_ARG(states != nullptr);
    for (uint32_t i = 0; i < num_layer; ++i) GGMS_CHECK_ARG(fanouts[i] > 0 && fanouts[i] < 50);
  }
  if (sample_type == GGMS_RANDOM_WALK)
    GGMS_CHECK_ARG(extra && extra->data && extra->random_walk_length > 0 && extra->num_random_walk > 0 && states);
  hipStream_t s = to_stream(stream);
  const BatchCaps c = caps_of(num_seeds, fanouts, num_layer);
  GGMS_CHECK_ARG(c.max_input[0] + c.max_edges[0] <= ht->n2o_size);
  GGMS_CHECK_ARG(c.max_input[0] + c.max_edges[0] < (1ull << 32) - 4); // indices and 2 + local id fit 32 bits
  if (sample_type == GGMS_WEIGHTED_KHOP_HASH_DEDUP) GGMS_CHECK_ARG((c.max_in_all + 1023) / 1024 * 256 <= num_states);
  if (sample_type == GGMS_KHOP2) { // unsharded CSR, mutated in place (dist_loops.cc:217-224)
    GGMS_CHECK_ARG(graph->num_part == 0 && graph->indptr && graph->indices);
    GGMS_CHECK_ARG(states && (c.max_in_all + 1023) / 1024 * 256 <= num_states);
  }
  if (sample_type == GGMS_KHOP3) {
    GGMS_CHECK_ARG(states && (c.max_in_all + 127) / 128 * 8 <= num_states);
    for (uint32_t i = 0; i < num_layer; ++i) GGMS_CHECK_ARG(fanouts[i] > 0 && fanouts[i] < 128);
  }

  const BatchLayout lay = layout_of(sample_type, num_seeds, fanouts, num_layer, c, extra);
  uint32_t *w = (uint32_t *)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  uint32_t *seed_local = w + lay.seed_local;
  uint32_t *samp_ws = w + lay.samp_ws;
  uint32_t *item_pos = w + lay.cand;
  unsigned long long *lost = reinterpret_cast<unsigned long long *>(w + lay.lost);
  // every ordered scan of the batch (seed offsets, owner flags) shares one control/descriptor area that is
  // cleared once here: descriptors are epoch-tagged, the control words re-arm themselves
  ScanArea scan{w + lay.scan, true};
  scan.chunk = w + lay.chunk;
  scan.tickets = scan.chunk + chunk_desc_words(); // chunk_desc_words() is a multiple of 4: the sets stay 16-byte aligned
  // every kernel of this batch reports a bound it hits into the BATCH's status word (behind the table's item counter;
  // zero since ggms_hashtable_init or the previous batch's last kernel), never into a word shared with the batches
  // in flight beside it (include/ggms.h, "Status words")
  scan.status = ht->num_items_dev + 1;

  GraphView g;
  if (!view_of(graph, g)) return GGMS_ERR_INVALID;
  // hash_table->Reset (dist_loops.cc:105): a new version stamp; the item count is zeroed by the prologue below
  if (ht->version >= 0x7ffffff0u) {
    int rc0 = ggms_hashtable_init(ht, stream);
    if (rc0 != GGMS_OK) return rc0;
  }
  ht->version += 1;
  // FillWithDupRevised(seeds), dist_loops.cc:110-111, + the local ids of the raw seeds (they may repeat): the
  // first layer's `col`.  Its insert kernel is the first kernel of the batch and carries the prologue:
  // scan-area clear, item count reset, num_dst of the first layer = |seeds| (dist_loops.cc:305).
  BatchPrologue pro{scan_align(scan.words), (uint32_t)(8 + 2 * lay.scan_tiles),
                    scan.chunk, (uint32_t)(chunk_desc_words() + (size_t)kTicketSets * kTicketWords), ht->num_items_dev,
                    counts_dev + 3 * (num_layer - 1) + 2};
  // direct table = batch mode of the dedup protocol (ggms_device.h): one index space for the whole batch, seeds first
  DedupInsert di{};
  di.cand = item_pos;
  di.lost = lost;
  di.batch = ht->direct ? 1u : 0u;
  di.base = 0;
  di.map.n = 1;
  di.map.base[0] = 0;
  di.map.arr[0] = seed_local;
  // Seeds the caller promises to be distinct (ggms_sample_extra_t.seeds_distinct), direct table: seed i is item i of the
  // unique list and its own local id -- no insert / ordered scan / look-up launches.  khop3 enters them inside the
  // first layer's launch (FirstLayer), every other sampler with one small launch (k_seed_enter).
  const bool distinct = extra && extra->seeds_distinct && ht->direct && num_seeds != 0;
  const bool fuse_seeds = distinct && sample_type == GGMS_KHOP3 && khop3_can_fuse_seeds(num_seeds) &&
                          c.max_edges[num_layer - 1] != 0;
  // khop0: its plan pass (both running sums of the first layer) reads the seeds anyway and enters them on the way
  const bool khop0_enters = distinct && sample_type == GGMS_KHOP0 && khop0_can_enter_seeds(num_seeds) &&
                            c.max_input[num_layer - 1] != 0;
  FirstLayer first_layer{};
  SeedEnter seed_enter{};
  int rc = GGMS_OK;
  if (distinct) {
    seed_local = nullptr;       // SrcMode: the first layer's `col` is the seed's position
    di.map.arr[0] = nullptr;    // IdxMap: segment 0 is the identity
    pro.items_are_seeds = 1;
    if (khop0_enters) {
      // as below: the prologue must leave alone the descriptors its own launch publishes (epoch-tagged; the rest of
      // the area is zeroed as usual)
      const size_t used = khop0_plan_desc_words(num_seeds);
      pro.num_zero = 8;
      pro.zero_words3 = pro.zero_words + 8 + used;
      pro.num_zero3 = (uint32_t)(2 * lay.scan_tiles > used ? 2 * lay.scan_tiles - used : 0);
      seed_enter = SeedEnter{(unsigned long long *)ht->o2n, ht->version, ht->n2o, pro};
    } else if (fuse_seeds) {
      // the prologue rides on the first layer's launch and must leave alone the tile descriptors that launch uses
      // (words [8, 8 + 2 tiles) of the area; they are epoch-tagged like every descriptor shared inside a batch)
      const size_t tiles0 = (num_seeds + 127) / 128;
      pro.num_zero = 8;
      pro.zero_words3 = pro.zero_words + 8 + 2 * tiles0;
      pro.num_zero3 = (uint32_t)(lay.scan_tiles > tiles0 ? 2 * (lay.scan_tiles - tiles0) : 0);
      first_layer.pro = pro;
      first_layer.n2o = ht->n2o;
    } else {
      rc = seed_enter_impl(ht, seeds, num_seeds, pro, s);
    }
  } else {
    rc = ht_fill_impl(ht, seeds, num_seeds, count_of(num_seeds), di, false, scan, nullptr, nullptr, s, seed_local, &pro,
                      kRestNow, pro.record_n);
  }
  if (rc != GGMS_OK) return rc;
  uint32_t next_base = (uint32_t)num_seeds;

  MapRestJobs jobs{};
  uint32_t num_jobs = 0;
  size_t job_items = 0;
  for (int i = (int)num_layer - 1; i >= 0; --i) {
    const bool first = (i == (int)num_layer - 1);
    const uint32_t *input = first ? seeds : ht->n2o;
    uint32_t *tmp_dst = w + lay.tmp_dst[i];
    const size_t n_max = c.max_input[i], e_max = c.max_edges[i];
    const Count n = first ? count_of(num_seeds) : count_of32(n_max, ht->num_items_dev);
    uint64_t *num_edge = counts_dev + 3 * i + 0;
    uint64_t *num_src = counts_dev + 3 * i + 1;                       // unique nodes after this layer (:304)
    uint64_t *next_dst = i > 0 ? counts_dev + 3 * (i - 1) + 2         // = next layer's frontier size (:305)
                               : counts_dev + 3 * num_layer;          // = number of input nodes
    // batch order on the shared RNG pool (and on khop2's CSR): only the sampler kernels are ordered
    if (first && extra && extra->rng_wait) GGMS_HIP(hipStreamWaitEvent(s, (hipEvent_t)extra->rng_wait, 0));
    if (i == 0 && extra && extra->heavy_wait) GGMS_HIP(hipStreamWaitEvent(s, (hipEvent_t)extra->heavy_wait, 0));
    // direct table + khop3: the sampler enters its output into the table itself (DedupInsert)
    di.base = next_base; // this layer's edges take the indices [base, base + e_max)
    di.w = (unsigned long long *)ht->o2n;
    di.version = ht->version;
    bool inserted = false;
    if (n_max == 0) {
      GGMS_HIP(hipMemsetAsync(num_edge, 0, sizeof(uint64_t), s));
    } else if (sample_type == GGMS_KHOP3) {
      inserted = ht->direct != 0 && e_max != 0;
      if (inserted) di.tag = next_dedup_tag();
      rc = sample_khop3_impl(g, input, n_max, n, (uint32_t)fanouts[i], col[i], tmp_dst, num_edge, (uint32_t *)states,
                             samp_ws, first ? seed_local : nullptr, 1, s, &scan, inserted ? &di : nullptr,
                             first && fuse_seeds ? &first_layer : nullptr);
    } else if (sample_type == GGMS_KHOP0) {
      inserted = ht->direct != 0 && e_max != 0; // khop0 enters its output where it produces it, too
      if (inserted) di.tag = next_dedup_tag();
      rc = sample_khop0_impl(g, input, n_max, n, (uint32_t)fanouts[i], col[i], tmp_dst, num_edge, samp_ws,
                             first ? seed_local : nullptr, 1, s, &scan, inserted ? &di : nullptr,
                             first && khop0_enters ? &seed_enter : nullptr);
    } else if (sample_type == GGMS_KHOP2) {
      // no fused insert: even with the four seeds of a lane in lock-step and their atomics issued together, the
      // returning atomics sit in the draw loop's dependency chain (measured on products: 0.45 -> 0.62 ms per step;
      // the separate insert launch of ht_fill_impl follows)
      rc = sample_khop2_impl(graph->indptr, const_cast<uint32_t *>(graph->indices), graph->num_node, input, n_max, n,
                             (uint32_t)fanouts[i], col[i], tmp_dst, num_edge, (uint32_t *)states, samp_ws,
                             first ? seed_local : nullptr, 1, s, &scan);
    } else if (sample_type == GGMS_KHOP1) {
      inserted = ht->direct != 0 && e_max != 0; // the weighted family enters its output in the compaction's emit
      if (inserted) di.tag = next_dedup_tag();
      rc = sample_weighted_impl(graph->indptr, graph->indices, nullptr, nullptr, input, n_max, n,
                                (uint32_t)fanouts[i], col[i], tmp_dst, num_edge, (uint32_t *)states, samp_ws,
                                first ? seed_local : nullptr, 1, s, &scan, graph->num_node, inserted ? &di : nullptr);
    } else if (sample_type == GGMS_WEIGHTED_KHOP_PREFIX) {
      inserted = ht->direct != 0 && e_max != 0;
      if (inserted) di.tag = next_dedup_tag();
      rc = sample_weighted_impl(graph->indptr, graph->indices, extra->prob_table, nullptr, input, n_max, n,
                                (uint32_t)fanouts[i], col[i], tmp_dst, num_edge, (uint32_t *)states, samp_ws,
                                first ? seed_local : nullptr, 1, s, &scan, graph->num_node, inserted ? &di : nullptr);
    } else if (sample_type == GGMS_WEIGHTED_KHOP_HASH_DEDUP) {
      inserted = ht->direct != 0 && e_max != 0; // a seed's picks are entered by its 16 lanes once the seed is done
      if (inserted) di.tag = next_dedup_tag();
      rc = sample_weighted_hash_dedup_impl(graph->indptr, graph->indices, extra->prob_table, extra->alias_table, input,
                                           n_max, n, (uint32_t)fanouts[i], col[i], tmp_dst, num_edge,
                                           (uint32_t *)states, samp_ws, first ? seed_local : nullptr, 1, s, &scan,
                                           inserted ? &di : nullptr);
    } else if (sample_type == GGMS_WEIGHTED_KHOP) {
      inserted = ht->direct != 0 && e_max != 0;
      if (inserted) di.tag = next_dedup_tag();
      rc = sample_weighted_impl(graph->indptr, graph->indices, extra->prob_table, extra->alias_table, input, n_max, n,
                                (uint32_t)fanouts[i], col[i], tmp_dst, num_edge, (uint32_t *)states, samp_ws,
                                first ? seed_local : nullptr, 1, s, &scan, graph->num_node, inserted ? &di : nullptr);
    } else { // random walk: fanout[i] = num_neighbor = K (operation.cc:174); enters its output like khop3
      inserted = ht->direct != 0 && e_max != 0;
      if (inserted) di.tag = next_dedup_tag();
      rc = sample_random_walk_impl(g, input, n_max, n, (uint32_t)extra->random_walk_length,
                                   extra->random_walk_restart_prob, (uint32_t)extra->num_random_walk,
                                   (uint32_t)fanouts[i], col[i], tmp_dst, extra->data[i], num_edge, (uint32_t *)states,
                                   samp_ws, first ? seed_local : nullptr, 1, s, &scan, inserted ? &di : nullptr);
    }
    if (rc != GGMS_OK) return rc;
    if (i == 0 && extra && extra->rng_done) GGMS_HIP(hipEventRecord((hipEvent_t)extra->rng_done, s));
    const Count ne = count_of(e_max, num_edge);
    if (e_max == 0) { // nothing can be sampled: the counts are the current table size
      hipLaunchKernelGGL(k_record, dim3(1), dim3(64), 0, s, num_src, count_of32(0, ht->num_items_dev));
      hipLaunchKernelGGL(k_record, dim3(1), dim3(64), 0, s, next_dst, count_of32(0, ht->num_items_dev));
      GGMS_LAUNCH_CHECK();
      continue;
    }
    // FillWithDuplicates (:279) + the dst half of GPUMapEdges (:296): row[i] = local id of every sampled neighbour
    // (direct table: the instances that do not own their key are resolved for all layers at once, below)
    rc = ht_fill_impl(ht, tmp_dst, e_max, ne, di, inserted, scan, num_src, next_dst, s, row[i], nullptr,
                      ht->direct ? kRestDefer : kRestNow, nullptr);
    if (rc != GGMS_OK) return rc;
    if (ht->direct) {
      di.map.base[di.map.n] = next_base; // later fills (and the final look-ups) find this layer's local ids in row[i]
      di.map.arr[di.map.n] = row[i];
      ++di.map.n;
      next_base += (uint32_t)e_max;
      jobs.row[num_jobs] = row[i];
      jobs.key[num_jobs] = tmp_dst;
      jobs.num[num_jobs] = num_edge;
      ++num_jobs;
      job_items = std::max(job_items, e_max);
    }
  }
  // the rest of GPUMapEdges' dst half for every layer + the batch's status word (counts_dev[3 L + 1])
  return launch_map_rest_all(ht, jobs, num_jobs, job_items, di.map, counts_dev + 3 * num_layer + 1, s);
}

} // extern "C"
