/*
 * ggms.h -- C ABI of the MI355X-native GGMS hot path (libggms_hip.so).
 *
 * Leaf operators: one entry point per reference leaf function, same argument
 * meaning, so that a maintainer of the reference can bind them where the CUDA
 * leaf is called today.  Citations are file:line relative to
 * /root/reference/samgraph/common/.
 *
 * Conventions
 *   - every pointer argument is DEVICE memory (or host memory mapped into the
 *     device: hipHostMalloc / hipHostRegister) unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; NULL = the null stream;
 *   - no entry point allocates, frees or synchronises: scratch comes from the
 *     caller (`ggms_*_workspace_bytes`), counts are written to device memory;
 *     the reference's per-kernel StreamSync (e.g. cuda_sampling_khop3.cu:269)
 *     is gone by design;
 *   - ids are uint32 (IdType, constant.h:28); 0xffffffff is kEmptyKey (:75);
 *   - return value: 0 on success, a negative ggms_status otherwise;
 *     ggms_last_error() describes the last failure of the calling thread.
 */
#ifndef GGMS_H
#define GGMS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t ggms_id_t;
typedef void *ggms_stream_t;
typedef void *ggms_event_t;   /* hipEvent_t */
#define GGMS_EMPTY_KEY 0xffffffffu

enum ggms_status {
  GGMS_OK = 0,
  GGMS_ERR_INVALID = -1, /* bad argument (null pointer, size out of range) */
  GGMS_ERR_HIP = -2,     /* a HIP runtime call or launch failed             */
  GGMS_ERR_NO_DEVICE = -3
};

/* DataType, common.h:38-46 (same integer codes) */
enum ggms_dtype {
  GGMS_F32 = 0,
  GGMS_F64 = 1,
  GGMS_F16 = 2,
  GGMS_U8 = 3,
  GGMS_I32 = 4,
  GGMS_I8 = 5,
  GGMS_I64 = 6
};

int ggms_abi_version(void);
const char *ggms_last_error(void);

/* Status words.  The reference CHECK-aborts when a device-side bound is hit (logging.cc:69-73); kernels here
 * cannot abort, so they OR a bit into a status word and return:
 *   GGMS_STATUS_SCAN_SPIN   an ordered scan's look-back gave up waiting for a predecessor tile (protocol error);
 *   GGMS_STATUS_TABLE_FULL  the hashed dedup table had no free bucket for a key (sized too small).
 * Results of a call that set a bit are invalid.  Two kinds of word:
 *   - every BATCH has its own: the second word behind its table's item counter (ggms_hashtable_t.num_items_dev[1],
 *     zeroed by ggms_hashtable_init).  Every kernel of a ggms_sample_batch call ORs into that word only, the
 *     batch's last kernel moves it into counts_dev[3 L + 1] (exchange with 0: the next batch on that table starts
 *     clean) -- so with several batches in flight a failure is reported by the batch it happened in, and only by
 *     it.  A non-zero word is also ORed into the device word below;
 *   - one per DEVICE, sticky, for the leaf operators (and as the record of failed batches): it stays set until
 *     ggms_device_status(clear != 0), which drains the device first (hipDeviceSynchronize: kernels on
 *     non-blocking streams included) and then zeroes it. */
#define GGMS_STATUS_SCAN_SPIN 1u
#define GGMS_STATUS_TABLE_FULL 2u
int ggms_device_status(uint32_t *status_host, int clear);
/* Rate probe for the bench report (no reference counterpart): num_requests random requests on a caller-owned table
 * of table_words 64-bit words -- GGMS_PROBE_ATOMIC: one returning 64-bit atomicMin each (what a dedup insert is),
 * GGMS_PROBE_LOAD: one 4-byte load each (what a neighbour read is), GGMS_PROBE_LOAD_ATOMIC: both (the sampler's
 * per-edge mix).  The caller times the launch; `salt` (use a decreasing sequence) changes the key set per launch and
 * keeps every atomic a real update.  sink: 4 device bytes. */
#define GGMS_PROBE_ATOMIC 0
#define GGMS_PROBE_LOAD 1
#define GGMS_PROBE_LOAD_ATOMIC 2
int ggms_fabric_probe(int kind, void *table, size_t table_words, size_t num_requests, uint32_t salt, void *sink,
                      ggms_stream_t stream);
/* Test aid: the next ordered scan of a direct-layout table fill (this process) starts with a poisoned ticket, so
 * that its look-back runs into its bound and sets GGMS_STATUS_SCAN_SPIN (the launch waits without helping itself).
 * One shot. */
void ggms_debug_poison_next_scan(void);
/* The ordered scans chain their tiles by look-back.  A look-back waits `polls` polls (default 2048, about a
 * millisecond) for a predecessor's word and then computes it from the scan's input itself, so that no workgroup's
 * progress depends on when or where another one runs (two processes or two batches sharing a GPU).  0: never wait
 * (tests: every look-back that finds a word missing takes the self-serve path); 0xffffffff: never help. */
void ggms_debug_set_scan_patience(uint32_t polls);
/* Test aid: in the next single-pass ordered scan of this process (the owner scan of a table fill, either layout) the
 * workgroup of tile / chunk 0 sleeps `sleeps` x s_sleep(127) (about 4 us each) before it reads anything -- with a small
 * patience the others compute its word themselves and the last one replaces the table's item count by the total before
 * that workgroup starts: the "owner arrives late" case of the self-serve look-back.  One shot. */
void ggms_debug_delay_next_scan(uint32_t sleeps);
/* Test aids (this process): force a slower, result-identical form of a kernel that default sizes rarely or never
 * reach.  value < 0 restores the default.  The library reads NO environment variable for any of this (the one it
 * reads is GGMS_EXTRACT_BLOCKS, the gather's grid cap).
 *   GGMS_DEBUG_KHOP0_DRAW_CAP     draws parked per khop0 launch (small: seeds whose draws do not fit are resolved in
 *                                 place by the generating lanes)
 *   GGMS_DEBUG_OWNER_SCAN_CHUNKS  workgroups of the chunked owner scan (small: chunks too long for registers are read
 *                                 twice)
 *   GGMS_DEBUG_OWNER_SCAN_TILES   != 0: the tile-chained owner scan (by default only beyond 65 M items per fill) */
#define GGMS_DEBUG_KHOP0_DRAW_CAP 0
#define GGMS_DEBUG_OWNER_SCAN_CHUNKS 1
#define GGMS_DEBUG_OWNER_SCAN_TILES 2
#define GGMS_DEBUG_NUM_KNOBS 3
void ggms_debug_set_knob(int knob, long long value);
size_t ggms_dtype_bytes(int dtype);

/* ---------------------------------------------------------------------------
 * Graph view.  num_part == 0: DeviceNormalGraph (cuda/dist_graph.h:160-180),
 * indptr/indices are one CSR.  num_part > 0: DeviceDistGraph (:114-158): node
 * v < num_cache_node lives in shard v % num_part at row v / num_part; every
 * other node in slot num_part (the whole CSR, normally pinned host memory).
 * part_indptr / part_indices are HOST arrays of num_part + 1 device pointers
 * (num_part <= GGMS_MAX_PARTS): the nine pointers of an 8-GPU node travel in the
 * kernels' arguments (SGPRs), where the reference's DeviceDistGraph keeps them in
 * a device array that every seed's lookup reads first (dist_graph.h:150-157);
 * `v % num_part`, `v / num_part` are a multiply-high by a launch constant.
 * ------------------------------------------------------------------------- */
#define GGMS_MAX_PARTS 8
typedef struct {
  const ggms_id_t *indptr;
  const ggms_id_t *indices;
  const ggms_id_t *const *part_indptr;
  const ggms_id_t *const *part_indices;
  uint32_t num_part;
  uint32_t num_cache_node;
  uint32_t num_node;
  uint32_t _pad;
} ggms_graph_t;

/* ---------------------------------------------------------------------------
 * XORWOW state pool -- GPURandomStates, cuda/cuda_random_states.cu:36-109.
 * One state is 6 x uint32 {d, v[5]} (24 B; cuRAND's Box-Muller fields are not
 * kept).  states[t] = curand_init(seed + t, 0, 0).
 * ------------------------------------------------------------------------- */
#define GGMS_RNG_STATE_BYTES 24
int ggms_random_states_init(void *states, size_t num_states, uint64_t seed,
                            ggms_stream_t stream);
/* sizing rule of cuda_random_states.cu:70-97 (host arithmetic only) */
size_t ggms_random_states_count(int sample_type, const size_t *fanout,
                                size_t num_fanout, size_t batch_size,
                                size_t num_random_walk);

/* ---------------------------------------------------------------------------
 * Neighbour samplers.  Output is the compact COO the reference produces:
 * (out_src = seed global id, out_dst = neighbour global id), seed order,
 * *num_out_dev (device, 64-bit like the reference's size_t) = number of edges.
 * out_src/out_dst must hold num_input * fanout entries.
 * ------------------------------------------------------------------------- */
/* SampleType, common/__init__.py:47-58 / common.h */
enum ggms_sample_type {
  GGMS_KHOP0 = 0,
  GGMS_KHOP1 = 1,
  GGMS_WEIGHTED_KHOP = 2,
  GGMS_RANDOM_WALK = 3,
  GGMS_WEIGHTED_KHOP_PREFIX = 4,
  GGMS_KHOP2 = 5,
  GGMS_WEIGHTED_KHOP_HASH_DEDUP = 6,
  GGMS_KHOP3 = 7
};

size_t ggms_sample_workspace_bytes(int sample_type, size_t num_input,
                                   size_t fanout);

/* GPUSampleKHop3<G>, cuda/cuda_sampling_khop3.cu:234-318 */
int ggms_sample_khop3(const ggms_graph_t *graph, const ggms_id_t *input,
                      size_t num_input, size_t fanout, ggms_id_t *out_src,
                      ggms_id_t *out_dst, uint64_t *num_out_dev, void *states,
                      size_t num_states, void *workspace,
                      size_t workspace_bytes, ggms_stream_t stream);

/* GPUSampleKHop0<G> (NEW_ALGO), cuda/cuda_sampling_khop0.cu:243-335 */
int ggms_sample_khop0(const ggms_graph_t *graph, const ggms_id_t *input,
                      size_t num_input, size_t fanout, ggms_id_t *out_src,
                      ggms_id_t *out_dst, uint64_t *num_out_dev,
                      void *workspace, size_t workspace_bytes,
                      ggms_stream_t stream);

/* GPUSampleWeightedKHopPrefix, cuda/cuda_sampling_weighted_khop_prefix.cu:145-246:
 * prob_prefix_table = per-list inclusive prefix sums of the edge weights; one
 * curand_uniform per task, binary search, then the weighted_khop sort/compaction.
 * Workspace and num_states as ggms_sample_weighted_khop. */
int ggms_sample_weighted_khop_prefix(const ggms_graph_t *graph,
                                     const float *prob_prefix_table,
                                     const ggms_id_t *input, size_t num_input,
                                     size_t fanout, ggms_id_t *out_src,
                                     ggms_id_t *out_dst, uint64_t *num_out_dev,
                                     void *states, size_t num_states,
                                     void *workspace, size_t workspace_bytes,
                                     ggms_stream_t stream);
/* GPUSampleWeightedKHopHashDedup, cuda/cuda_sampling_weighted_khop_hash_dedup.cu:
 * 196-283: alias-method candidates until `fanout` (< 50) distinct ids per seed;
 * stream map of khop2 (num_states >= ceil(num_input/1024)*256).  Two bounds the
 * reference lacks (it would not terminate): a seed that has drawn 65536
 * candidates takes every further one, and a table probe gives up after 64 steps
 * (full table) and takes the candidate.
 * Workspace: ggms_sample_workspace_bytes. */
int ggms_sample_weighted_khop_hash_dedup(const ggms_graph_t *graph,
                                         const float *prob_table,
                                         const ggms_id_t *alias_table,
                                         const ggms_id_t *input,
                                         size_t num_input, size_t fanout,
                                         ggms_id_t *out_src, ggms_id_t *out_dst,
                                         uint64_t *num_out_dev, void *states,
                                         size_t num_states, void *workspace,
                                         size_t workspace_bytes,
                                         ggms_stream_t stream);

/* GPUSampleKHop1, cuda/cuda_sampling_khop1.cu:130-236: uniform WITH replacement,
 * stable order by src id, an edge equal to its successor dropped.  Unsharded graphs
 * only (dist_loops.cc:167-168).  Workspace: ggms_sample_weighted_workspace_bytes.
 * num_states >= min(tasks, roundup256(min(tasks, 512K))), tasks = num_input*fanout. */
int ggms_sample_khop1(const ggms_graph_t *graph, const ggms_id_t *input,
                      size_t num_input, size_t fanout, ggms_id_t *out_src,
                      ggms_id_t *out_dst, uint64_t *num_out_dev, void *states,
                      size_t num_states, void *workspace,
                      size_t workspace_bytes, ggms_stream_t stream);

/* GPUSampleKHop2 (ORIGIN_KHOP2), cuda/cuda_sampling_khop2.cu:196-262.  In-place
 * partial Fisher-Yates: graph->indices is PERMUTED (the reference const_casts it,
 * cuda_loops.cc:163); unsharded graphs only (dist_loops.cc:219); the seeds of one
 * call must be distinct.  num_states >= ceil(num_input/1024)*256 (khop2.cu:57). */
int ggms_sample_khop2(const ggms_graph_t *graph, const ggms_id_t *input,
                      size_t num_input, size_t fanout, ggms_id_t *out_src,
                      ggms_id_t *out_dst, uint64_t *num_out_dev, void *states,
                      size_t num_states, void *workspace,
                      size_t workspace_bytes, ggms_stream_t stream);

/* GPUSampleWeightedKHop, cuda/cuda_sampling_weighted_khop.cu:132-238 (alias
 * method, with replacement, stable order by src, adjacent duplicates dropped).
 * prob_table f32[E], alias_table u32[E] hold GLOBAL node ids
 * (utility/data-process/toolkit/weight/create_alias_table.cc:154-155).
 * DeviceNormalGraph only, like the reference (dist/dist_loops.cc:171-172). */
size_t ggms_sample_weighted_workspace_bytes(size_t num_input, size_t fanout);
int ggms_sample_weighted_khop(const ggms_graph_t *graph, const float *prob_table,
                              const ggms_id_t *alias_table,
                              const ggms_id_t *input, size_t num_input,
                              size_t fanout, ggms_id_t *out_src,
                              ggms_id_t *out_dst, uint64_t *num_out_dev,
                              void *states, size_t num_states, void *workspace,
                              size_t workspace_bytes, ggms_stream_t stream);

/* GPUSampleRandomWalk + FrequencyHashmap::GetTopK, cuda/cuda_sampling_random_walk.cu:116-165,
 * cuda/cuda_frequency_hashmap.cu:643-841: per seed, num_walk walks of walk_length steps with restart;
 * output = the K most visited nodes per seed (count descending, ties by first visit), seeds in input order;
 * out_data = visit count.  out_* hold num_input * K entries.  Up to 128 visits per seed
 * (walk_length * num_walk; PinSAGE's defaults: 12) are ranked in LDS, more in the visit scratch itself (slower). */
size_t ggms_sample_random_walk_workspace_bytes(size_t num_input, size_t walk_length,
                                               size_t num_walk, size_t K);
size_t ggms_random_walk_num_states(size_t num_input, size_t num_walk); /* cuda_random_states.cu:48-60 */
int ggms_sample_random_walk(const ggms_graph_t *graph, const ggms_id_t *input,
                            size_t num_input, size_t walk_length,
                            double restart_prob, size_t num_walk, size_t K,
                            ggms_id_t *out_src, ggms_id_t *out_dst,
                            ggms_id_t *out_data, uint64_t *num_out_dev,
                            void *states, size_t num_states, void *workspace,
                            size_t workspace_bytes, ggms_stream_t stream);

/* ---------------------------------------------------------------------------
 * Ordered hash table -- OrderedHashTable, cuda/cuda_hashtable.h:103-153,
 * cuda/cuda_hashtable.cu:699-1064.  Dedup with contiguous, prefix-stable local
 * ids; n2o[0 .. num_items) is the unique list.  Canonical order among new
 * keys: first occurrence in the input (DESIGN.md "canonical semantics").
 * The caller owns the three buffers; the struct is plain data.
 * ------------------------------------------------------------------------- */
typedef struct {
  void *o2n;               /* hashed: o2n_size buckets of 16 B; direct: o2n_size words of 8 B */
  ggms_id_t *n2o;          /* n2o_size ids                                   */
  uint32_t *num_items_dev; /* TWO device words: [0] item counter, [1] status
                              word of the batch that runs on this table       */
  uint64_t o2n_size;       /* hashed: power of two; direct: >= number of node ids */
  uint64_t n2o_size;
  uint32_t version;        /* bumped by ggms_hashtable_reset                 */
  uint32_t direct;         /* 0: hashed layout (reference sizing, TableSize);
                              1: direct-mapped, one 8-byte word per node id --
                              no probing, one atomic per insert; needs
                              8 B x num_node of HBM              */
} ggms_hashtable_t;

#define GGMS_HT_BUCKET_BYTES 16
/* TableSize(num, scale = 2), cuda_hashtable.cu:146-149 */
size_t ggms_hashtable_num_buckets(size_t capacity);
/* constructor body, cuda_hashtable.cu:699-729: fills buckets with 0xff */
int ggms_hashtable_init(ggms_hashtable_t *ht, ggms_stream_t stream);
/* Reset, cuda_hashtable.cu:739-742: O(1) version bump (+ counter clear) */
int ggms_hashtable_reset(ggms_hashtable_t *ht, ggms_stream_t stream);
size_t ggms_hashtable_workspace_bytes(size_t num_input);
/* FillWithDuplicates :744-837 and FillWithDupRevised :850-912.  If
 * unique_out != NULL the n2o prefix is also copied there (what
 * FillWithDuplicates returns).  *ht->num_items_dev is updated on device. */
int ggms_hashtable_fill_with_duplicates(ggms_hashtable_t *ht,
                                        const ggms_id_t *input,
                                        size_t num_input,
                                        ggms_id_t *unique_out, void *workspace,
                                        size_t workspace_bytes,
                                        ggms_stream_t stream);
/* GPUMapEdges, cuda/cuda_mapping.cu:68-81 */
int ggms_map_edges(const ggms_hashtable_t *ht, const ggms_id_t *global_src,
                   ggms_id_t *new_src, const ggms_id_t *global_dst,
                   ggms_id_t *new_dst, size_t num_edges, ggms_stream_t stream);

/* ---------------------------------------------------------------------------
 * One mini-batch of k-layer sampling -- DoGPUSample, dist/dist_loops.cc:62-368
 * (cuda/cuda_loops.cc:54-292): Reset + FillWithDupRevised(seeds), then for
 * layer i = L-1 .. 0 {sample, FillWithDuplicates, GPUMapEdges}.  Enqueued on
 * `stream` without any host round trip; all sizes are left on the device:
 *   counts_dev[3*i + 0] = num_edge(i)   row[i]/col[i] hold that many entries
 *   counts_dev[3*i + 1] = num_src(i)    (= unique nodes after layer i)
 *   counts_dev[3*i + 2] = num_dst(i)    (= size of layer i's frontier)
 *   counts_dev[3*L]     = number of input nodes; the list is ht->n2o
 *   counts_dev[3*L + 1] = status word of THIS batch (0 = ok; "Status words" above)
 * row[i] = local id of the sampled neighbour, col[i] = local id of the seed
 * (TrainGraph, dist_loops.cc:303-322).  row/col are HOST arrays of L device
 * pointers with the capacities ggms_sample_batch_capacity reports; fanouts is
 * a host array indexed by layer id.
 * The table is CONSUMED by the batch: with the direct layout its words hold the
 * index of each key's first occurrence, not local ids (the ids are in row / n2o),
 * so ggms_map_edges on it is meaningless until the next fill after a reset.
 * ------------------------------------------------------------------------- */
/* per-sample-type extras of ggms_sample_batch (NULL for khop0/khop3) */
typedef struct {
  const float *prob_table;        /* weighted_khop[_hash_dedup]: dataset->prob_table (engine.cc:372-384);
                                     weighted_khop_prefix: dataset->prob_prefix_table, alias_table NULL */
  const ggms_id_t *alias_table;   /*                dataset->alias_table                      */
  size_t random_walk_length;      /* random_walk: RunConfig::random_walk_length ...           */
  double random_walk_restart_prob;
  size_t num_random_walk;
  ggms_id_t *const *data;         /* random_walk: HOST array of L device pointers, visit counts
                                     (TrainGraph::data, dist_loops.cc:314-319)                */
  /* Several batches in flight (one stream, table and workspace each) must still consume the
   * shared RNG pool -- and, for khop2, permute the CSR -- in batch order, as the reference's
   * one-batch-at-a-time loop does.  The sampler kernels of this batch wait for `rng_wait` and
   * `rng_done` is recorded right after the last of them; everything else of the batch (table
   * fills, scans, id mapping) is free to overlap with the neighbouring batches.  NULL = none. */
  ggms_event_t rng_wait;
  ggms_event_t rng_done;
  /* Optional: the sampler launch of the LAST layer processed (layer 0, normally by far the largest: most of the
   * batch's neighbour loads and table atomics) waits for this event.  A caller that gathers the previous batch's
   * feature rows on another stream passes that gather's completion event: the memory fabric then serves the
   * gather and the fabric-heaviest sampler kernel one after the other instead of time-slicing them;
   * the smaller layers still overlap the gather.  NULL = no wait.  Results do not depend on it. */
  ggms_event_t heavy_wait;
  /* != 0: the caller promises that the seeds are pairwise distinct -- a slice of a shuffled train set is
   * (cuda_shuffler.cc:89-110); the padded tail of an aligned epoch (dist_shuffler_aligned.cc:46-63) and raw leaf
   * calls may not be.  FillWithDupRevised(seeds) (dist_loops.cc:105-111) then has nothing to decide: local id =
   * position, the unique list starts with the seeds as they are.  With the direct table layout the batch skips the
   * seeds' insert / ordered scan / look-up launches: khop3 enters the seeds into the table inside the first layer's
   * launch (their indices [0, S) win every atomicMin whenever they arrive; a neighbour instance that got there
   * first is told so through `lost`, like any other beaten candidate), the other samplers use one small launch.
   * Results are identical to the general path for distinct seeds; with duplicated seeds they are undefined. */
  uint32_t seeds_distinct;
  uint32_t _pad;
} ggms_sample_extra_t;

/* events for the ordering above (thin hipEvent_t handles, timing disabled) */
int ggms_event_create(ggms_event_t *event);
int ggms_event_destroy(ggms_event_t event);

int ggms_sample_batch_capacity(size_t num_seeds, const size_t *fanouts,
                               uint32_t num_layer, size_t *max_input,
                               size_t *max_edges, size_t *max_unique);
size_t ggms_sample_batch_workspace_bytes(int sample_type, size_t num_seeds,
                                         const size_t *fanouts,
                                         uint32_t num_layer,
                                         const ggms_sample_extra_t *extra);
int ggms_sample_batch(int sample_type, const ggms_graph_t *graph,
                      const ggms_id_t *seeds, size_t num_seeds,
                      const size_t *fanouts, uint32_t num_layer,
                      ggms_hashtable_t *ht, void *states, size_t num_states,
                      ggms_id_t *const *row, ggms_id_t *const *col,
                      uint64_t *counts_dev, const ggms_sample_extra_t *extra,
                      void *workspace, size_t workspace_bytes,
                      ggms_stream_t stream);

/* ---------------------------------------------------------------------------
 * Feature extract -- GPUExtract, cuda/cuda_extraction.cu:74-117:
 * dst[i, :] = src[index[i], :].  src may be device or device-mapped host
 * memory (gpu_extract zero-copy path, dist_loops.cc:585-634).
 * ------------------------------------------------------------------------- */
int ggms_extract(void *dst, const void *src, const ggms_id_t *index,
                 size_t num_index, size_t dim, int dtype,
                 ggms_stream_t stream);

/* GPUMockExtract, cuda/cuda_extraction.cu:51-70,119-160 (SAMGRAPH_EMPTY_FEAT = k: the feature table is a
 * 2^k-row stand-in, engine.cc:198-235): dst[i, :] = src[index[i] & (2^mock_bits - 1), :].
 * ggms_gather_scatter_masked is the general form (device count, optional scatter), as the miss extract of the
 * cache manager uses it (cuda_cache_manager_host.cc:47-48). */
int ggms_mock_extract(void *dst, const void *src, const ggms_id_t *index,
                      size_t num_index, size_t dim, int dtype,
                      uint32_t mock_bits, ggms_stream_t stream);
int ggms_gather_scatter_masked(void *out, const void *src,
                               const ggms_id_t *src_index,
                               const ggms_id_t *dst_index, size_t num,
                               const uint64_t *num_dev, size_t dim, int dtype,
                               uint32_t src_row_mask, ggms_stream_t stream);

/* ---------------------------------------------------------------------------
 * Feature cache -- GPUCacheManager, cuda/cuda_cache_manager_device.cu.
 * table[node] = cache slot or kEmptyKey.
 * ------------------------------------------------------------------------- */
/* presample ranking (dist/pre_sampler.cc:79-111): freq[nodes[i]] += 1 (the reference copies the batch's
 * input nodes to the host and counts there); num may be overridden by a device count. */
int ggms_count_nodes(uint32_t *freq, const ggms_id_t *nodes, size_t num_nodes,
                     const uint64_t *num_nodes_dev, ggms_stream_t stream);
size_t ggms_cache_index_workspace_bytes(size_t num_nodes);
/* GetMissCacheIndex :355-441 (kernels :40-169): stable split of `nodes` into
 * miss (src = global id, dst = output row) and hit (src = slot, dst = row). */
int ggms_get_miss_cache_index(const ggms_id_t *table, const ggms_id_t *nodes,
                              size_t num_nodes, ggms_id_t *miss_src_index,
                              ggms_id_t *miss_dst_index,
                              uint64_t *num_miss_dev,
                              ggms_id_t *cache_src_index,
                              ggms_id_t *cache_dst_index,
                              uint64_t *num_cache_dev, void *workspace,
                              size_t workspace_bytes, ggms_stream_t stream);
/* the same with the batch size left on the device (num_nodes = upper bound that sizes the launch and the
 * workspace; num_nodes_dev NULL = num_nodes is exact): the split can be enqueued right behind the sampler. */
int ggms_get_miss_cache_index_dev(const ggms_id_t *table, const ggms_id_t *nodes,
                                  size_t num_nodes, const uint64_t *num_nodes_dev,
                                  ggms_id_t *miss_src_index, ggms_id_t *miss_dst_index,
                                  uint64_t *num_miss_dev, ggms_id_t *cache_src_index,
                                  ggms_id_t *cache_dst_index, uint64_t *num_cache_dev,
                                  void *workspace, size_t workspace_bytes,
                                  ggms_stream_t stream);
/* combine_cache_data :254-275, extract_miss_data :233-252, combine_miss_data
 * :209-231: out[dst_index[i], :] = src[src_index[i], :]; a NULL index means
 * the identity.  num may be overridden by a device count (num_dev != NULL;
 * `num` is then the upper bound used to size the launch). */
int ggms_gather_scatter(void *out, const void *src, const ggms_id_t *src_index,
                        const ggms_id_t *dst_index, size_t num,
                        const uint64_t *num_dev, size_t dim, int dtype,
                        ggms_stream_t stream);
/* combine_cache_data_for_partition :277-299 with DeviceDistFeature
 * (cuda/dist_graph.h:182-212): slot s lives in parts[s % num_part] at row
 * s / num_part.  parts: HOST array of num_part (<= GGMS_MAX_PARTS) base
 * pointers (local HBM, peer HBM mapped with hipIpcOpenMemHandle, or mapped host
 * memory); they are passed to the kernel by value. */
int ggms_gather_scatter_partition(void *out, const void *const *parts,
                                  uint32_t num_part,
                                  const ggms_id_t *src_index,
                                  const ggms_id_t *dst_index, size_t num,
                                  const uint64_t *num_dev, size_t dim,
                                  int dtype, ggms_stream_t stream);
/* One-pass replacement of GetMissCacheIndex + GPUExtractMissData +
 * CombineCacheData (dist_loops.cc:1209-1285): out[i,:] = table[nodes[i]] ==
 * kEmptyKey ? host_feat[nodes[i],:] : parts[slot % P][slot / P,:].
 * num_part == 0 -> single cache array parts[0].  Also counts misses.  parts: HOST array, as above.
 * table == NULL: the whole feature table is cached and kept in NODE order (slot = node id):
 * out[i,:] = parts[node % P][node / P,:] with no table read and no miss tier -- the layout of a full cache
 * is not observable through the reference's interface, and the gather loses one dependent random read per row.
 * The same convention holds for ggms_owner_histogram. */
int ggms_extract_cached(void *out, const ggms_id_t *nodes, size_t num_nodes,
                        const uint64_t *num_nodes_dev, const ggms_id_t *table,
                        const void *const *parts, uint32_t num_part,
                        const void *host_feat, size_t dim, int dtype,
                        uint64_t *num_miss_dev, ggms_stream_t stream);

/* Every tier of the store behind one gather (GGMS across GPUs with hot-row replication):
 *   slot = table ? table[node] : node            (table NULL: full cache, slot = node id)
 *   slot == kEmptyKey      host_feat[node]       pinned host DRAM, zero-copy        (GPUExtractMissData :573-625)
 *   slot <  num_replica    replica[slot]         this GPU's copy of the hottest rows (what PartitionSolver's
 *                                                replica placement buys on NVLink, dist_graph.cu:40-222)
 *   else s = slot - num_replica                  parts[s % num_part][s / num_part]: local HBM or a peer's shard
 *                                                over xGMI (combine_cache_data_for_partition :277-299)
 * tier_rows_dev (optional, 4 zeroed uint64): rows served by {host, remote shard, local shard, replica} are
 * ADDED to it -- count_local_cache (:171-207) generalised. */
typedef struct {
  const ggms_id_t *table;
  const void *replica;
  uint64_t num_replica;
  const void *const *parts; /* HOST array of num_part (<= GGMS_MAX_PARTS) base pointers */
  uint32_t num_part;
  uint32_t my_part;
  const void *host_feat;
  uint32_t host_row_mask; /* 0 = none; else host row = node & mask (mock table, SAMGRAPH_EMPTY_FEAT) */
  uint32_t _pad;
} ggms_feature_tiers_t;
int ggms_extract_tiered(void *out, const ggms_id_t *nodes, size_t num_nodes,
                        const uint64_t *num_nodes_dev,
                        const ggms_feature_tiers_t *tiers, size_t dim, int dtype,
                        uint64_t *tier_rows_dev, ggms_stream_t stream);


/* ---------------------------------------------------------------------------
 * Launch timer: a row gather's OWN start / end timestamps, with no packet of their own on the stream.
 * The reference times its extract with a host timer around a stream sync (dist_loops.cc:1276-1281,
 * kLogL1CopyTime); a pipelined caller has to use events instead, and every hipEventRecord / hipStreamWaitEvent
 * is a barrier packet the command processor works through between two kernels of the stream -- 6-7 us each on
 * MI355X, all of it dead time on the stream that bounds the step (profiles/r05_ab_extract_stream.txt).  A timer's
 * two events ride ON the dispatch packet of the next row-gather launch (hipExtLaunchKernel): the kernel's own
 * start and end timestamps, and its end event is what another stream waits for ("the slot's rows are out").
 *   arm(t):      the next launch of ggms_extract* / ggms_gather_scatter* / ggms_mock_extract issued by THIS thread
 *                carries the timer (thread-local, consumed by that one launch; a call that launches nothing --
 *                zero rows -- leaves it armed)
 *   wait(t, s):  stream s waits for the timed launch (no-op if the timer never rode a launch)
 *   elapsed_us:  blocks until the launch has finished; GGMS_ERR_INVALID if the timer never rode a launch
 *   span_us:     see below
 * A timer belongs to the thread that arms it until its launch has been issued: destroy it from that thread, or after.
 * ------------------------------------------------------------------------- */
typedef struct ggms_launch_timer ggms_launch_timer_t;
int ggms_launch_timer_create(ggms_launch_timer_t **timer);
int ggms_launch_timer_destroy(ggms_launch_timer_t *timer);
int ggms_launch_timer_arm(ggms_launch_timer_t *timer);
int ggms_launch_timer_wait(ggms_launch_timer_t *timer, ggms_stream_t stream);
int ggms_launch_timer_elapsed_us(ggms_launch_timer_t *timer, double *us);
/* from the start of `first`'s launch to the end of `last`'s (both finished: blocks for them): with launches that may
 * overlap on two streams, sum of durations / span = how many are in flight on average */
int ggms_launch_timer_span_us(ggms_launch_timer_t *first, ggms_launch_timer_t *last, double *us);

/* ---------------------------------------------------------------------------
 * GGMS shards across processes (one process per GPU).
 *
 * Publishing a shard: DistGraph::_Barrier/IPC exchange, cuda/dist_graph.cu:228-272
 * (cudaIpcGetMemHandle / cudaIpcOpenMemHandle).  A shard is an allocation of its
 * own (ggms_device_alloc), so the handle maps exactly the shard.  `handle` is a
 * GGMS_IPC_HANDLE_BYTES buffer that may travel through any byte channel
 * (shared memory, torch.distributed.all_gather_object, ...).
 * ------------------------------------------------------------------------- */
#define GGMS_IPC_HANDLE_BYTES 64
/* Size an allocation must have to be safely opened by a peer process.  Measured on ROCm 7.2 / MI355X
 * (tools/ipc_probe3.py): hipIpcOpenMemHandle never returns when the exporter's allocation size has bit 31 set
 * (size mod 2^32 >= 2^31: 3000 MiB, 12000 MiB, 20000 MiB, 28000 MiB hang; 1500, 6000, 17408, 20480, 28672 MiB
 * open in a millisecond).  Such sizes are rounded up to the next multiple of 4 GiB; ggms_device_alloc applies it. */
size_t ggms_ipc_safe_bytes(size_t bytes); /* host arithmetic only */
int ggms_device_alloc(void **ptr, size_t bytes);
int ggms_device_free(void *ptr);
int ggms_ipc_export(const void *ptr, void *handle);
int ggms_ipc_import(const void *handle, void **ptr);   /* peer HBM, read in-kernel over xGMI */
int ggms_ipc_release(void *ptr);

/* ---------------------------------------------------------------------------
 * Link / topology probe -- PartitionSolver::DetectTopo, DetectTopo_child, LoadTopoFromFile
 * (cuda/dist_graph.cu:684-726, 779-884, 886-938): P2P reachability of every GPU pair + one timed 128-MiB copy per
 * reachable pair, kept in the text file the reference's solver reads ("GPU Count" / "P2P Matrix" / "Bandwidth Matrix";
 * a file written by either implementation loads in the other).
 *   ggms_detect_topology   for ONE process that sees every GPU (the engine's forked probe child): it creates a context
 *                          on every device -- never call it from a bench rank or an engine worker.
 *                          can_access[i][j] = hipDeviceCanAccessPeer(i, j), 1 on the diagonal; copy_GBps[i][j] = GB/s of
 *                          a copy INTO device i FROM device j (diagonal: local copy, 2 x bytes, as the reference counts it).
 *   ggms_peer_access       one pair, from a process that must not touch the peer device (no context is created on it).
 *   ggms_link_probe_copy / _gather   what a rank of the one-process-per-GPU deployment measures on the hipIpc mappings
 *                          it already holds: a timed copy out of a mapping, and the product's own gather kernel
 *                          (ggms_gather_scatter_partition) reading random rows of it -- one part = one peer alone, all
 *                          parts = the rank's whole inbound xGMI.  Unlike the operators above these two SYNCHRONISE the
 *                          stream (they return a rate).
 * ------------------------------------------------------------------------- */
#define GGMS_TOPO_MAX_DEVICE 16
typedef struct {
  int32_t num_device;
  int32_t _pad;
  int32_t can_access[GGMS_TOPO_MAX_DEVICE][GGMS_TOPO_MAX_DEVICE];
  double copy_GBps[GGMS_TOPO_MAX_DEVICE][GGMS_TOPO_MAX_DEVICE];
} ggms_topology_t;
int ggms_device_count(int *count);
int ggms_peer_access(int device, int peer, int *can_access);
int ggms_detect_topology(ggms_topology_t *topo, size_t probe_bytes /* 0 = 128 MiB */, int reps);
int ggms_topology_write_host(const ggms_topology_t *topo, const char *path, const char *device_order);
int ggms_topology_read_host(ggms_topology_t *topo, const char *path);
/* with_kernel 0: hipMemcpyAsync; k = 1 .. 16: a plain 16-B-per-lane streaming copy kernel, k workgroups per CU (bytes,
 * dst, src multiples of 16) */
int ggms_link_probe_copy(void *dst, const void *src, size_t bytes, int reps, int with_kernel,
                         double *GBps_host, ggms_stream_t stream);
/* out: num_rows x row_bytes; parts: HOST array of num_part (<= GGMS_MAX_PARTS) base pointers, each holding
 * rows_per_part rows; index_ws: num_rows ids of device scratch; row_bytes a multiple of 4. */
int ggms_link_probe_gather(void *out, const void *const *parts, uint32_t num_part, size_t rows_per_part,
                           size_t row_bytes, size_t num_rows, uint32_t seed, int reps, ggms_id_t *index_ws,
                           double *GBps_host, ggms_stream_t stream);

/* The exchange form of the remote gather: instead of dereferencing
 * peer pointers inside the gather kernel (ggms_extract_cached with num_part > 0),
 * a batch's rows are requested from their owners with one all-to-all of row ids
 * and returned with one all-to-all of rows.  These two leaves split the batch by
 * owner: slot s = table[node] lives on shard s % num_part at row s / num_part
 * (cuda_cache_manager_host.cc:187-221); an uncached node (slot kEmptyKey) goes to
 * bucket num_part and keeps its node id (host tier).
 *   ggms_owner_histogram  slots_out[i] = table[nodes[i]]; counts_dev[p] += |bucket p|
 *                         (counts_dev: num_part + 1 zeroed uint64)
 *   ggms_owner_bucket     cursor_dev[p] = start of bucket p on entry (exclusive
 *                         prefix of the counts), its end on return;
 *                         bucket_row[k] = row id to ask the owner for (node id in
 *                         the host bucket), bucket_pos[k] = row of the batch output
 *                         it fills.  Order inside a bucket is unspecified. */
int ggms_owner_histogram(const ggms_id_t *table, const ggms_id_t *nodes,
                         size_t num_nodes, const uint64_t *num_nodes_dev,
                         uint32_t num_part, ggms_id_t *slots_out,
                         uint64_t *counts_dev, ggms_stream_t stream);
int ggms_owner_bucket(const ggms_id_t *slots, const ggms_id_t *nodes,
                      size_t num_nodes, const uint64_t *num_nodes_dev,
                      uint32_t num_part, uint64_t *cursor_dev,
                      ggms_id_t *bucket_row, ggms_id_t *bucket_pos,
                      ggms_stream_t stream);

/* ---------------------------------------------------------------------------
 * Dataset tools (HOST arrays, no device work): the per-edge tables of the weighted samplers.
 *   ggms_build_alias_table_host        utility/data-process/toolkit/weight/create_alias_table.cc:105-170:
 *       per neighbour list Vose's alias method in float arithmetic; prob_table[e] = acceptance probability of
 *       slot e, alias_table[e] = GLOBAL node id of the donor neighbour (0 where the slot accepts with
 *       probability 1).  -> prob_table.bin / alias_table.bin
 *   ggms_build_prob_prefix_table_host  create_prob_prefix_table.cc:94-123: running float sum of the weights per
 *       list.  -> prob_prefix_table.bin
 * `weights` is one float per edge (the reference draws them inside the tool; here they are an input).
 * ------------------------------------------------------------------------- */
int ggms_build_alias_table_host(const ggms_id_t *indptr, const ggms_id_t *indices,
                                size_t num_node, const float *weights,
                                float *prob_table, ggms_id_t *alias_table,
                                int num_threads);
int ggms_build_prob_prefix_table_host(const ggms_id_t *indptr, size_t num_node,
                                      const float *weights,
                                      float *prob_prefix_table, int num_threads);

#ifdef __cplusplus
}
#endif
#endif /* GGMS_H */
