/*
 * ggms_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's GGMS hot path: neighbour
 * samplers, ordered dedup / remap, cache hit/miss split and feature row
 * gather.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (xgnn_amd/) never does.
 *
 * Citations are file:line relative to /root/reference/samgraph/common/.
 *
 * Parity status
 *   - CPU-engine leaves (orc_cpu_*):  pinned against the reference's own CPU
 *     objects compiled into oracle/_ref (see oracle/Makefile) through
 *     the tests/golden npz files.
 *   - GPU-engine semantics (orc_sample_khop3 / khop0 / weighted / random walk /
 *     top-k, XORWOW):  **parity unpinned** -- the reference has no golden
 *     vectors for them and its CUDA engine cannot be built here.  They follow
 *     the .cu text under the canonical choices listed in DESIGN.md
 *     (lock-step group draw, highest-j-wins reservoir, first-occurrence dedup).
 *     cuRAND XORWOW constants are restated from the published curand_kernel.h
 *     (CUDA 11.7) and are unverified against a CUDA device.
 */
#ifndef GGMS_ORACLE_H
#define GGMS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint32_t orc_id_t;             /* IdType, constant.h:28 */
#define ORC_EMPTY_KEY 0xffffffffu      /* Constant::kEmptyKey, constant.h:75 */

/* ---- cuRAND XORWOW (third-party: CUDA 11.7 curand_kernel.h) ------------- */
typedef struct {
  uint32_t d;
  uint32_t v[5];
} orc_xorwow_t;

void orc_xorwow_init(orc_xorwow_t *st, uint64_t seed); /* curand_init(seed,0,0) */
uint32_t orc_xorwow_next(orc_xorwow_t *st);            /* curand()              */
float orc_xorwow_uniform(orc_xorwow_t *st);            /* curand_uniform()      */
double orc_xorwow_uniform_double(orc_xorwow_t *st);    /* curand_uniform_double */
void orc_xorwow_draws(orc_xorwow_t *st, size_t n, uint32_t *out);
uint32_t orc_xorwow_fold(orc_xorwow_t *st, size_t n);  /* acc = rotl(acc,1) ^ draw */
void orc_xorwow_uniforms(orc_xorwow_t *st, size_t n, float *out);
void orc_xorwow_uniform_doubles(orc_xorwow_t *st, size_t n, double *out);
/* cuda_random_states.cu:36-46: states[t] = curand_init(seed + t, 0, 0) */
void orc_random_states_init(orc_xorwow_t *states, size_t num, uint64_t seed);

/* ---- libstdc++ generators / distributions (third party: GCC 11 libstdc++) */
typedef struct {
  uint32_t mt[624];
  int idx;
} orc_mt19937_t;
void orc_mt19937_seed(orc_mt19937_t *g, uint32_t seed);
uint32_t orc_mt19937_next(orc_mt19937_t *g);
/* std::uniform_int_distribution<uint32_t>(lo,hi)(std::mt19937&) */
uint32_t orc_mt19937_uniform_u32(orc_mt19937_t *g, uint32_t lo, uint32_t hi);
/* std::shuffle(first, first+n, std::mt19937&) on uint32 data */
void orc_mt19937_shuffle_u32(orc_mt19937_t *g, uint32_t *data, size_t n);
/* std::minstd_rand0 (== std::default_random_engine) */
typedef struct {
  uint64_t x;
} orc_minstd0_t;
void orc_minstd0_seed(orc_minstd0_t *g, uint64_t seed);
uint64_t orc_minstd0_next(orc_minstd0_t *g);
/* std::uniform_int_distribution<size_t>(lo,hi)(std::default_random_engine&) */
uint64_t orc_minstd0_uniform_u64(orc_minstd0_t *g, uint64_t lo, uint64_t hi);

/* ---- shufflers ---------------------------------------------------------- */
/* Fisher-Yates of dist_shuffler_aligned.cc:89-113 / cpu_shuffler.cc:68-90 */
void orc_shuffle_minstd0(uint32_t *data, size_t n, uint64_t seed);
/* DistAlignedShuffler ctor padding, dist_shuffler_aligned.cc:46-56.
 * out must hold round_up(n, num_worker) entries; returns that count. */
size_t orc_aligned_pad(const uint32_t *train, size_t n, size_t num_worker,
                       uint32_t *out);
/* GetBatch offset/size, dist_shuffler_aligned.cc:123-146 */
void orc_aligned_batch_range(size_t num_local_data, size_t batch_size,
                             size_t epoch, size_t local_step, size_t *offset,
                             size_t *size);

/* ---- CPU engine leaves (arch0) ------------------------------------------ */
/* cpu_random.cc:26-30: one process-wide (per thread) default-seeded mt19937 */
void orc_cpu_random_reset(void);
uint32_t orc_cpu_random_id(uint32_t lo, uint32_t hi);
/* cpu_sampling_khop0.cc:29-83 */
void orc_cpu_sample_khop0(const orc_id_t *indptr, const orc_id_t *indices,
                          const orc_id_t *input, size_t num_input,
                          orc_id_t *out_src, orc_id_t *out_dst,
                          size_t *num_out, size_t fanout, int num_threads);
/* cpu_sampling_khop2.cc:29-76; MUTATES indices */
void orc_cpu_sample_khop2(const orc_id_t *indptr, orc_id_t *indices,
                          const orc_id_t *input, size_t num_input,
                          orc_id_t *out_src, orc_id_t *out_dst,
                          size_t *num_out, size_t fanout, int num_threads);
/* cpu_extraction.cc:31-90 (row copy; dim*esize = row_bytes) */
void orc_extract(void *dst, const void *src, const orc_id_t *index,
                 size_t num_index, size_t row_bytes, int num_threads);

/* ---- GPU-engine samplers (semantics of the .cu files) ------------------- */
/* common.cc:488-497 */
size_t orc_predict_num_nodes(size_t batch_size, const size_t *fanout,
                             size_t num_fanout_to_comp);
/* cuda_sampling_khop3.cu:76-146 + count/compact :148-230 */
void orc_sample_khop3(const orc_id_t *indptr, const orc_id_t *indices,
                      const orc_id_t *input, size_t num_input, size_t fanout,
                      orc_xorwow_t *states, size_t num_states,
                      orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out);
/* cuda_sampling_weighted_khop_prefix.cu:41-246 (prob_prefix_table: inclusive prefix sums per neighbour list) */
void orc_sample_weighted_khop_prefix(const orc_id_t *indptr, const orc_id_t *indices,
                                     const float *prob_prefix_table,
                                     const orc_id_t *input, size_t num_input, size_t fanout,
                                     orc_xorwow_t *states, size_t num_states,
                                     orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out);
/* cuda_sampling_weighted_khop_hash_dedup.cu:41-283 */
#define ORC_HASH_DEDUP_MAX_TRIES 65536
#define ORC_HASH_DEDUP_MAX_PROBES 64
void orc_sample_weighted_khop_hash_dedup(const orc_id_t *indptr, const orc_id_t *indices,
                                         const float *prob_table, const orc_id_t *alias_table,
                                         const orc_id_t *input, size_t num_input, size_t fanout,
                                         orc_xorwow_t *states, size_t num_states,
                                         orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out);
/* cuda_sampling_khop1.cu:42-127,130-236 */
void orc_sample_khop1(const orc_id_t *indptr, const orc_id_t *indices,
                      const orc_id_t *input, size_t num_input, size_t fanout,
                      orc_xorwow_t *states, size_t num_states,
                      orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out);
/* cuda_sampling_khop2.cu:46-95 (ORIGIN_KHOP2) + count/compact :102-194; MUTATES indices */
void orc_sample_khop2(const orc_id_t *indptr, orc_id_t *indices,
                      const orc_id_t *input, size_t num_input, size_t fanout,
                      orc_xorwow_t *states, size_t num_states,
                      orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out);
/* cuda_sampling_khop0.cu:102-153 (NEW_ALGO) + count/compact :157-239 */
void orc_sample_khop0(const orc_id_t *indptr, const orc_id_t *indices,
                      const orc_id_t *input, size_t num_input, size_t fanout,
                      orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out);
/* cuda_sampling_weighted_khop.cu:41-128,132-238 */
void orc_sample_weighted_khop(const orc_id_t *indptr, const orc_id_t *indices,
                              const float *prob_table,
                              const orc_id_t *alias_table,
                              const orc_id_t *input, size_t num_input,
                              size_t fanout, orc_xorwow_t *states,
                              size_t num_states, orc_id_t *out_src,
                              orc_id_t *out_dst, size_t *num_out);
/* cuda_sampling_random_walk.cu:43-112 + cuda_frequency_hashmap.cu GetTopK */
void orc_sample_random_walk(const orc_id_t *indptr, const orc_id_t *indices,
                            const orc_id_t *input, size_t num_input,
                            size_t walk_length, double restart_prob,
                            size_t num_walk, size_t K, orc_xorwow_t *states,
                            size_t num_states, orc_id_t *out_src,
                            orc_id_t *out_dst, orc_id_t *out_data,
                            size_t *num_out);
/* raw walk output before top-k (tmp_src/tmp_dst of random_walk.cu:128-147) */
void orc_random_walk_raw(const orc_id_t *indptr, const orc_id_t *indices,
                         const orc_id_t *input, size_t num_input,
                         size_t walk_length, double restart_prob,
                         size_t num_walk, orc_xorwow_t *states,
                         size_t num_states, orc_id_t *tmp_src,
                         orc_id_t *tmp_dst);

/* ---- ordered hash table (cuda_hashtable.cu / cpu_hashtable2.cc) --------- */
typedef struct orc_hashtable orc_hashtable_t;
/* dataset tools: create_alias_table.cc:105-170, create_prob_prefix_table.cc:94-123 */
void orc_create_alias_table(const orc_id_t *indptr, const orc_id_t *indices, size_t num_node, const float *weights,
                            float *prob_table, orc_id_t *alias_table);
void orc_create_prob_prefix_table(const orc_id_t *indptr, size_t num_node, const float *weights, float *prefix);

/* CPUHashTable2 with its OpenMP loops (cpu/cpu_hashtable2.cc:35-191); one thread == orc_ht_* above */
typedef struct orc_cpu_ht2 orc_cpu_ht2_t;
orc_cpu_ht2_t *orc_cpu_ht2_create(size_t num_node, int threads);
void orc_cpu_ht2_destroy(orc_cpu_ht2_t *ht);
void orc_cpu_ht2_reset(orc_cpu_ht2_t *ht, int threads);
size_t orc_cpu_ht2_populate(orc_cpu_ht2_t *ht, const orc_id_t *input, size_t num_input, int threads);
const orc_id_t *orc_cpu_ht2_unique(const orc_cpu_ht2_t *ht);
void orc_cpu_ht2_map_edges(const orc_cpu_ht2_t *ht, const orc_id_t *src, const orc_id_t *dst, size_t len,
                           orc_id_t *new_src, orc_id_t *new_dst, int threads);
orc_hashtable_t *orc_ht_create(size_t max_node_id_plus1, size_t capacity);
void orc_ht_destroy(orc_hashtable_t *ht);
void orc_ht_reset(orc_hashtable_t *ht);        /* cuda_hashtable.cu:739-742 */
/* FillWithDupRevised / FillWithDuplicates (first occurrence wins; new local
 * ids in first-occurrence order after the existing ones).  Returns NumItems.*/
size_t orc_ht_fill_with_duplicates(orc_hashtable_t *ht, const orc_id_t *input,
                                   size_t num_input);
size_t orc_ht_num_items(const orc_hashtable_t *ht);
const orc_id_t *orc_ht_unique(const orc_hashtable_t *ht); /* n2o prefix */
/* cuda_mapping.cu:49-79 / cpu_hashtable2.cc:170-181 */
void orc_ht_map_edges(const orc_hashtable_t *ht, const orc_id_t *src,
                      const orc_id_t *dst, size_t num_edges, orc_id_t *new_src,
                      orc_id_t *new_dst);

/* ---- multi-layer sample loop (dist_loops.cc:62-368, cpu_loops.cc:55-192) */
enum { ORC_KHOP0 = 0, ORC_KHOP1 = 1, ORC_WEIGHTED_KHOP = 2, ORC_WEIGHTED_KHOP_PREFIX = 4, ORC_KHOP2 = 5,
       ORC_WEIGHTED_KHOP_HASH_DEDUP = 6, ORC_RANDOM_WALK = 3, ORC_KHOP3 = 7, ORC_CPU_KHOP0 = 100 };
typedef struct {
  const float *prob_table;
  const orc_id_t *alias_table;
  size_t walk_length;
  double restart_prob;
  size_t num_walk;
} orc_sample_extra_t;
typedef struct {
  size_t num_layer;
  size_t *num_src, *num_dst, *num_edge; /* per layer, index = layer id    */
  orc_id_t **row, **col;                /* row = nbr local, col = seed local */
  orc_id_t **data;                      /* random walk: visit counts (else NULL entries) */
  orc_id_t *input_nodes;                /* final unique list              */
  size_t num_input_nodes;
} orc_sample_result_t;
orc_sample_result_t *orc_do_sample(int sample_type, const orc_id_t *indptr,
                                   const orc_id_t *indices, size_t num_node,
                                   const orc_id_t *seeds, size_t num_seeds,
                                   const size_t *fanouts, size_t num_layer,
                                   orc_xorwow_t *states, size_t num_states);
orc_sample_result_t *orc_do_sample_ex(int sample_type, const orc_id_t *indptr,
                                      const orc_id_t *indices, size_t num_node,
                                      const orc_id_t *seeds, size_t num_seeds,
                                      const size_t *fanouts, size_t num_layer,
                                      orc_xorwow_t *states, size_t num_states,
                                      const orc_sample_extra_t *extra);
void orc_sample_result_free(orc_sample_result_t *r);

/* ---- feature cache (cuda_cache_manager_*.{cu,cc}) ----------------------- */
/* cuda_cache_manager_host.cc:96-107 (plain) / :164-229 (partition: prefix
 * shuffled with std::mt19937(num_cached)).  rank_out receives the (possibly
 * shuffled) rank list, table[num_nodes] the id -> slot map. */
void orc_cache_build(const orc_id_t *rank_nodes, size_t num_nodes,
                     size_t num_cached, int partition_shuffle,
                     orc_id_t *rank_out, orc_id_t *table);
/* cuda_cache_manager_device.cu:40-169,355-441 */
void orc_get_miss_cache_index(const orc_id_t *table, const orc_id_t *nodes,
                              size_t num_nodes, orc_id_t *miss_src,
                              orc_id_t *miss_dst, size_t *num_miss,
                              orc_id_t *hit_src, orc_id_t *hit_dst,
                              size_t *num_hit);
/* combine_cache_data / combine_miss_data / extract_miss_data :209-275 */
void orc_gather_scatter(void *out, const void *src, const orc_id_t *src_index,
                        const orc_id_t *dst_index, size_t n, size_t row_bytes);
/* combine_cache_data_for_partition :277-299 with DeviceDistFeature
 * (dist_graph.h:182-212): slot -> parts[slot % P] row slot / P */
void orc_gather_scatter_partition(void *out, const void *const *parts,
                                  size_t num_part, const orc_id_t *src_index,
                                  const orc_id_t *dst_index, size_t n,
                                  size_t row_bytes);

/* ---- GGMS sharding (dist_graph.cu) -------------------------------------- */
/* _DatasetPartition :228-272.  Pass NULL outputs to query sizes. */
void orc_partition_graph(const orc_id_t *indptr, const orc_id_t *indices,
                         orc_id_t part_id, orc_id_t num_part,
                         orc_id_t num_part_node, orc_id_t *part_indptr,
                         orc_id_t *part_indices, size_t *indptr_size,
                         size_t *indices_size);
/* _PartitionFeature :493-521 : rows rank[i], i = part, part+P, ... */
size_t orc_partition_feature(const void *feat, size_t row_bytes,
                             const orc_id_t *rank_nodes, orc_id_t num_cache,
                             orc_id_t part_id, orc_id_t num_part, void *out);
/* dist_engine.cc:223-232: first v with indptr[v] >= num_edge*percentage */
orc_id_t orc_num_cache_node(const orc_id_t *indptr, orc_id_t num_node,
                            double percentage);

#ifdef __cplusplus
}
#endif
#endif /* GGMS_ORACLE_H */
