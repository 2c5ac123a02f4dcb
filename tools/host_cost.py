#!/usr/bin/env python3
"""Host-side cost of one native training step (config 2): time to ENQUEUE n steps from an idle stream, n small enough that the
HIP queue never fills -- the Python + ctypes + 9 hipLaunchKernel calls, without any GPU back-pressure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from hydra_gnn_amd import workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = HeterogeneousNetwork(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3,
                           dropout=0.25).to(dev)
net.train()
batch = workloads.config2_batch(32).to(dev)
y = batch["rooms"].y
step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25)
for _ in range(50):
    step(batch, y)
torch.cuda.synchronize()
for n in (5, 10, 20, 40):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(batch, y)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, (t1 - t0) / n)
    print(f"n = {n:3d}: host enqueue {1e6 * best:6.1f} us/step   (wall incl. drain {1e6 * (t2 - t0) / n:6.1f} us/step)")
