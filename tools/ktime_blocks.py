#!/usr/bin/env python3
"""Life of every workgroup of one agg_proj_fwd launch (profiling build: make KTIME=1).

    HMP_LIB=hydra-gnn_amd/hydra_gnn_amd/libhydra_mp_kt.so python tools/ktime_blocks.py

2-layer model on the config-2 batch (one agg_proj_fwd launch per step): start / duration of each workgroup relative to the first
start, grouped by XCD (block id mod 8) and by node type (objects: 16-row tiles first, then the rooms' 8-row tiles)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
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
    for _ in range(30):
        step(batch, y)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (C.c_ulonglong * 2048)()
    fn = lib.hmp_debug_ktime_agg_blocks
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    n_obj = (batch["objects"].x.size(0) + 15) // 16
    n_room = (batch["rooms"].x.size(0) + 7) // 8  # rooms (average in-degree > 8) are cut into tiles of 8 rows
    nb = n_obj + n_room
    a = np.array(list(buf), dtype=np.int64).reshape(-1, 2)[:nb]
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0) / 100.0
    dur = (a[:, 1] - a[:, 0]) / 100.0
    end = (a[:, 1] - t0) / 100.0
    print(f"{nb} workgroups ({n_obj} object tiles, {n_room} room tiles); launch span {end.max():.2f} us")
    print(f"start: min {start.min():.2f} median {np.median(start):.2f} max {start.max():.2f} us")
    for name, sl in (("objects", slice(0, n_obj)), ("rooms", slice(n_obj, nb))):
        d = dur[sl]
        print(f"{name:8s} duration: min {d.min():.2f} p25 {np.percentile(d, 25):.2f} median {np.median(d):.2f} p75 {np.percentile(d, 75):.2f} "
              f"p95 {np.percentile(d, 95):.2f} max {d.max():.2f} us")
    for x in range(8):
        idx = np.arange(nb)[np.arange(nb) % 8 == x]
        print(f"XCD {x}: start median {np.median(start[idx]):.2f}  duration median {np.median(dur[idx]):.2f} max {dur[idx].max():.2f}  end max {end[idx].max():.2f}")
    order = np.argsort(-end)[:8]
    print("last to finish:", [(int(b), f"start {start[b]:.2f}", f"dur {dur[b]:.2f}") for b in order])
    # phase stamps of one object tile and one room tile (KT(0): tile start, KT(3..7): inside agg_row of the tile's first row group,
    # KT(1): gather done and tile in LDS, KT(2): projection stored)
    sel = lib.hmp_debug_ktime_agg_select
    sel.argtypes = [C.c_int]
    get = lib.hmp_debug_ktime_agg
    get.argtypes = [C.POINTER(C.c_ulonglong)]
    for name, blk in (("object tile 0", 0), ("room tile", n_obj + 2)):
        assert sel(blk) == 0
        for _ in range(3):
            step(batch, y)
        torch.cuda.synchronize()
        b64 = (C.c_ulonglong * 64)()
        assert get(b64) == 0
        v = list(b64)
        slots = [0, 3, 4, 5, 6, 7, 1, 2]
        print(f"{name:14s}", " ".join(f"[{i}]+{(v[i] - v[0]) / 100.0:5.2f}" for i in slots))
    sel(0)


if __name__ == "__main__":
    main()
