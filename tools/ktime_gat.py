#!/usr/bin/env python3
"""In-kernel phase times of gat_bwd1_kernel on config 3 (KTIME build), block 0 / thread 0 of the LAST launch (= layer 0)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from hydra_gnn_amd import _lib, workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = HeterogeneousNetwork(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GAT", GAT_hidden_dims=[128, 128],
                           GAT_heads=[4, 4, 4], GAT_concats=[True, True, False], dropout=0.25).to(dev)
net.train()
batch = workloads.config3_batch(64).to(dev)
y = batch["rooms"].y
step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25)
for _ in range(10):
    step(batch, y)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 64)()
fn = lib.hmp_debug_ktime_gat
fn.argtypes = [C.POINTER(C.c_ulonglong)]
fn.restype = C.c_int
assert fn(buf) == 0
v = list(buf)
base = v[0]
print("gat_bwd1 (layer 0, block 0):", " ".join(f"[{i}]+{(v[i] - base) / 100.0:.2f}" for i in range(0, 10) if v[i] >= base))
