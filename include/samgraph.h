/*
 * samgraph.h -- engine-level C ABI: the `samgraph_*` surface the reference's
 * Python packages bind (ctypes in samgraph/common/__init__.py:279-537, pybind11
 * in samgraph/torch/adapter.cc:195-206), exported by the same libggms_hip.so.
 *
 * Same names, argument meaning and error behaviour as
 * /root/reference/samgraph/common/operation.h:30-115 (citations below are
 * relative to /root/reference/samgraph/): configuration is a string map,
 * failures print and abort() the process (logging.cc:69-73), the engine is a
 * process-wide singleton with one "current batch".
 *
 * What differs, on purpose:
 *   - the nine pybind11 functions that returned torch::Tensor (adapter.cc:62-193)
 *     are plain C here: they fill a samgraph_tensor_t {pointer, shape, dtype,
 *     device}; the Python side wraps it zero-copy (__cuda_array_interface__).
 *     No torch type crosses the boundary;
 *   - batch buffers are reference counted explicitly (samgraph_batch_retain /
 *     _release) instead of through a captured shared_ptr (adapter.cc:70-73);
 *   - extra config keys (all optional): "seed" (RNG + shuffler seed; default =
 *     wall clock like the reference), "hash_table" = "direct" | "hashed".
 */
#ifndef SAMGRAPH_ABI_H
#define SAMGRAPH_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- configuration / lifecycle: common/operation.h:30-43, 96-113 --------- */
void samgraph_config(const char **config_keys, const char **config_values,
                     const size_t num_config_items);   /* operation.cc:49-62  */
void samgraph_init(void);                               /* :328-335 (single process: arch1) */
void samgraph_start(void);                              /* :337-345 */
void samgraph_shutdown(void);                           /* :392-398 */
void samgraph_data_init(void);                          /* :509-533 (parent, before fork)   */
void samgraph_sample_init(int worker_id, const char *ctx);  /* :535-540 */
void samgraph_train_init(int worker_id, const char *ctx);   /* :549-554 */
void samgraph_extract_start(int count);                     /* :556-559 */
int samgraph_wait_one_child(void);                          /* :573-584 */
void samgraph_forward_barrier(void);                        /* :507     */
/* Entry points of deployments this build does not carry (arch9 unified-memory sampling, arch5 switcher;
 * operation.cc:541-547,561-571): exported so that the reference's loader binds, they abort with a message
 * like any unsupported configuration (logging.cc:69-73). */
void samgraph_um_sample_init(int num_workers);
void samgraph_switch_init(int worker_id, const char *ctx, double cache_percentage);

/* ---- queries: operation.h:45-63 ------------------------------------------ */
size_t samgraph_num_epoch(void);
size_t samgraph_steps_per_epoch(void);
size_t samgraph_num_local_step(void);
size_t samgraph_num_class(void);
size_t samgraph_feat_dim(void);
uint64_t samgraph_get_next_batch(void);                 /* :366-378 */
void samgraph_sample_once(void);                        /* :380     */
size_t samgraph_get_graph_num_src(uint64_t key, int graph_id);
size_t samgraph_get_graph_num_dst(uint64_t key, int graph_id);
size_t samgraph_get_graph_num_edge(uint64_t key, int graph_id);

/* ---- profiler / trace: operation.h:65-94; item codes = common/profiler.h:30-163
 * (mirrored by name in xgnn_amd/common.py) -------------------------------- */
void samgraph_log_step(uint64_t epoch, uint64_t step, int item, double val);
void samgraph_log_step_by_key(uint64_t key, int item, double val);
void samgraph_log_step_add(uint64_t epoch, uint64_t step, int item, double val);
void samgraph_log_epoch_add(uint64_t epoch, int item, double val);
double samgraph_get_log_init_value(int item);
double samgraph_get_log_step_value(uint64_t epoch, uint64_t step, int item);
double samgraph_get_log_step_value_by_key(uint64_t key, int item);
double samgraph_get_log_epoch_value(uint64_t epoch, int item);
void samgraph_report_init(void);
void samgraph_report_step(uint64_t epoch, uint64_t step);
void samgraph_report_step_average(uint64_t epoch, uint64_t step);
void samgraph_report_epoch(uint64_t epoch);
void samgraph_report_epoch_average(uint64_t epoch);
void samgraph_report_node_access(void);
void samgraph_trace_step_begin(uint64_t key, int item, uint64_t ts);
void samgraph_trace_step_end(uint64_t key, int item, uint64_t ts);
void samgraph_trace_step_begin_now(uint64_t key, int item);
void samgraph_trace_step_end_now(uint64_t key, int item);
void samgraph_dump_trace(void);

/* ---- tensor hand-off: torch/adapter.cc:62-193 ---------------------------- */
typedef struct {
  void *data;
  int64_t shape[2];
  int32_t ndim;
  int32_t dtype;       /* ggms_dtype / DataType code (common/common.h:38-46) */
  int32_t device_type; /* 0 = host, 2 = GPU (DeviceType, common.h:48)       */
  int32_t device_id;
} samgraph_tensor_t;

/* each checks key == current batch key like CHECK_EQ(key, graph_batch->key) (adapter.cc:68) */
void samgraph_get_graph_feat(uint64_t key, samgraph_tensor_t *out);              /* :62-77   */
void samgraph_get_graph_label(uint64_t key, samgraph_tensor_t *out);             /* :79-92   */
void samgraph_get_graph_row(uint64_t key, int layer_idx, samgraph_tensor_t *out);/* :94-106  */
void samgraph_get_graph_col(uint64_t key, int layer_idx, samgraph_tensor_t *out);/* :108-120 */
void samgraph_get_graph_data(uint64_t key, int layer_idx, samgraph_tensor_t *out);/* :122-134 */
void samgraph_get_dataset_feat(samgraph_tensor_t *out);                          /* :136-152 */
void samgraph_get_dataset_label(samgraph_tensor_t *out);                         /* :154-167 */
void samgraph_get_graph_input_nodes(uint64_t key, samgraph_tensor_t *out);       /* :169-180 */
void samgraph_get_graph_output_nodes(uint64_t key, samgraph_tensor_t *out);      /* :182-193 */
/* a batch's buffers stay valid while its retain count is > 0 */
void samgraph_batch_retain(uint64_t key);
void samgraph_batch_release(uint64_t key);

#ifdef __cplusplus
}
#endif
#endif /* SAMGRAPH_ABI_H */
