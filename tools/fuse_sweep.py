#!/usr/bin/env python3
"""Crossover of the two launch sequences on the headline model: config-2 network (3-layer HeteroConv(SAGE),
hidden 64, fp32, dropout 0.25) at batch sizes 32 .. 2048 (2048 = the batch size of every config/mp3d/*.yaml), each with the
small-batch sequence pinned (HMP_FUSE=1), the stand-alone sequence pinned (HMP_FUSE=0) and the automatic choice.

    python tools/fuse_sweep.py [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from hydra_gnn_amd import workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402

KW = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3, dropout=0.25)


def measure(batch, mode, steps):
    if mode is None:
        os.environ.pop("HMP_FUSE", None)
    else:
        os.environ["HMP_FUSE"] = mode
    torch.manual_seed(0)
    net = HeterogeneousNetwork(**KW).to("cuda:0")
    net.train()
    y = batch["rooms"].y
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=1, use_graph=False)
    for _ in range(10):
        step(batch, y)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(batch, y)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    assert net.native().read_state()[1] == 0
    return best * 1e3


def main():
    rows = []
    for B in (32, 64, 128, 256, 512, 1024, 2048):
        batch = workloads.config2_batch(B).to("cuda:0")
        nodes = sum(batch[t].x.size(0) for t in ("objects", "rooms"))
        steps = max(20, 4000 // B)
        r = {"batch": B, "nodes": nodes, "edges": int(sum(batch[e].edge_index.size(1) for e in batch.edge_types))}
        for name, mode in (("fused_ms", "1"), ("standalone_ms", "0"), ("auto_ms", None)):
            r[name] = round(measure(batch, mode, steps), 4)
        r["graphs_per_s_auto"] = round(B / (r["auto_ms"] * 1e-3))
        rows.append(r)
        print(r, flush=True)
    os.environ.pop("HMP_FUSE", None)
    if len(sys.argv) > 1:
        json.dump({"what": __doc__.strip().split("\n\n")[0], "rows": rows}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
