from xgnn_amd.torch import *  # noqa: F401,F403
from xgnn_amd.torch import (config, init, start, num_class, feat_dim, num_epoch, steps_per_epoch,  # noqa: F401
                            get_next_batch, get_graph_num_src, get_graph_num_dst, shutdown, sample_once, log_step,
                            log_step_add, log_epoch_add, get_log_init_value, get_log_step_value, get_log_epoch_value,
                            report_init, report_step, report_step_average, report_epoch, report_epoch_average,
                            report_node_access, trace_step_begin, trace_step_end, trace_step_begin_now,
                            trace_step_end_now, dump_trace, forward_barrier, wait_one_child, log_step_by_key,
                            get_log_step_value_by_key, data_init, sample_init, train_init, extract_start,
                            num_local_step, um_sample_init, switch_init, get_graph_feat, get_graph_label, get_graph_row, get_graph_col,
                            get_graph_data, get_dgl_blocks, get_dgl_blocks_with_weights, get_dataset_feat,
                            get_dataset_label, get_graph_input_nodes, get_graph_output_nodes, load_subtensor,
                            notify_sampler_ready, wait_for_sampler_ready, get_graph_coo)
