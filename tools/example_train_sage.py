"""GraphSAGE (mean aggregator) trained on batches from the samgraph_* engine, in plain PyTorch-ROCm.

The reference's example scripts (example/samgraph/train_graphsage.py) build DGL blocks from the same tensors
(sam.get_dgl_blocks); DGL is not part of this image, so this example consumes the COO the engine hands over
(sam.get_graph_coo: row = local id of the sampled neighbour, col = local id of the node it was sampled for) with
index_add_.  Everything up to the model -- config keys, init, sample_once / get_next_batch, get_graph_feat /
label -- is the reference's own loop.

    python tools/example_train_sage.py <dataset_dir> [--epochs 2] [--batch-size 1024] [--fanout 10 5] [--sample-type khop3]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import samgraph.torch as sam  # noqa: E402


class SageLayer(torch.nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.w_self = torch.nn.Linear(d_in, d_out)
        self.w_nbr = torch.nn.Linear(d_in, d_out, bias=False)

    def forward(self, h, row, col, num_dst):
        # nodes are numbered so that the num_dst nodes a layer produces come first (seeds first, prefix-stable ids)
        agg = torch.zeros((num_dst, h.shape[1]), dtype=h.dtype, device=h.device)
        deg = torch.zeros(num_dst, dtype=h.dtype, device=h.device)
        agg.index_add_(0, col, h[row])
        deg.index_add_(0, col, torch.ones_like(col, dtype=h.dtype))
        return self.w_self(h[:num_dst]) + self.w_nbr(agg / deg.clamp(min=1).unsqueeze(1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--fanout", type=int, nargs="+", default=[10, 5])
    ap.add_argument("--sample-type", default="khop3")
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    L = len(args.fanout)
    sam.config({"dataset_path": args.dataset, "_arch": sam.builtin_archs["arch1"]["arch"],
                "_sample_type": sam.sample_types[args.sample_type], "batch_size": args.batch_size,
                "num_epoch": args.epochs, "_cache_policy": sam.cache_policies["degree"], "cache_percentage": 0.0,
                "max_sampling_jobs": 10, "max_copying_jobs": 1, "omp_thread_num": 4, "num_layer": L,
                "num_hidden": args.hidden, "lr": 0.003, "dropout": 0.0, "num_fanout": L, "fanout": args.fanout,
                "sampler_ctx": "cuda:0", "trainer_ctx": "cuda:0", "seed": args.seed})
    sam.init()
    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    dims = [sam.feat_dim()] + [args.hidden] * (L - 1) + [sam.num_class()]
    layers = torch.nn.ModuleList([SageLayer(dims[i], dims[i + 1]) for i in range(L)]).to(dev)
    opt = torch.optim.Adam(layers.parameters(), lr=0.003)
    steps = sam.steps_per_epoch()
    for epoch in range(args.epochs):
        t0, total, correct, loss_sum = time.time(), 0, 0, 0.0
        for step in range(steps):
            sam.sample_once()
            key = sam.get_next_batch()
            h = sam.get_graph_feat(key) / 65536.0  # the synthetic features are integers in [0, 65535]
            label = sam.get_graph_label(key)
            coo = sam.get_graph_coo(key, L)  # layer 0 = outermost hop (largest frontier)
            for i in range(L):
                row, col, num_src, num_dst = coo[i]
                assert h.shape[0] == num_src
                h = layers[i](h, row.long(), col.long(), num_dst)
                if i + 1 < L:
                    h = torch.relu(h)
            loss = torch.nn.functional.cross_entropy(h, label)
            opt.zero_grad()
            loss.backward()
            opt.step()
            total += label.numel()
            correct += int((h.argmax(1) == label).sum())
            loss_sum += float(loss) * label.numel()
        print(f"epoch {epoch}: {steps} steps, loss {loss_sum / total:.4f}, train acc {correct / total:.3f}, "
              f"{time.time() - t0:.2f} s, sampled {sam.get_log_epoch_value(epoch, sam.kLogEpochNumSample):.0f} edges",
              flush=True)
    sam.shutdown()


if __name__ == "__main__":
    main()
