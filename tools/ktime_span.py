#!/usr/bin/env python3
"""Span of one launch over all its workgroups vs the life of its longest workgroup (profiling build: make KTIME=1).

    HMP_LIB=hydra-gnn_amd/hydra_gnn_amd/libhydra_mp_kt.so python tools/ktime_span.py

A 2-layer model on the config-2 batch has exactly one agg_proj_fwd and one agg_bwd_dx launch per step: the stamps of every
workgroup's thread 0 give first start -> last end (the launch as the stream sees it, minus dispatch) and the longest workgroup."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from hydra_gnn_amd import _lib, workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = HeterogeneousNetwork(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64,
                               num_layers=2, dropout=0.25).to(dev)
    net.train()
    batch = workloads.config2_batch(32).to(dev)
    y = batch["rooms"].y
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, use_graph=False)
    for _ in range(20):
        step(batch, y)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (C.c_ulonglong * 64)()
    for base in (40, 48):
        buf[base] = 2 ** 64 - 1
    setter, getter = lib.hmp_debug_ktime_agg_set, lib.hmp_debug_ktime_agg
    setter.argtypes = getter.argtypes = [C.POINTER(C.c_ulonglong)]
    assert setter(buf) == 0
    step(batch, y)
    torch.cuda.synchronize()
    assert getter(buf) == 0
    for name, base in (("agg_proj_fwd (layer 0)", 40), ("agg_bwd_dx (layer 1)", 48)):
        print(f"{name:24s} first start -> last end {(buf[base + 1] - buf[base]) / 100.0:6.2f} us   longest workgroup: entry 0 (objects) "
              f"{buf[base + 2] / 100.0:6.2f} us, entry 1 (rooms) {buf[base + 3] / 100.0:6.2f} us")


if __name__ == "__main__":
    main()
