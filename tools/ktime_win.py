#!/usr/bin/env python3
"""Phase times of the LDS sliding-window aggregation (profiling build: make -C hydra-gnn_amd/csrc KTIME=1).

    HMP_LIB=hydra-gnn_amd/hydra_gnn_amd/libhydra_mp_kt.so python tools/ktime_win.py [n_objects]

Workgroup 0 / thread 0 of the LAST forward launch: accumulated wall time (us) of phase A (requests for the next chunks), phase
B+C (the chunk's rows), phase D (staged registers -> LDS), the barrier, and the whole loop."""
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
    n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    os.environ["HMP_BF16_ALL"] = "1"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = HeterogeneousNetwork(input_dim_dict={"objects": 256, "rooms": 256}, output_dim=26, conv_block="GraphSAGE", hidden_dim=256,
                               num_layers=3, dropout=0.25).to(dev)
    net.train()
    net.native().set_compute("bf16")
    g = workloads.big_hetero_graph(n_obj=n_obj, n_rooms=max(n_obj // 100, 1)).to(dev)
    y = g["rooms"].y
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, use_graph=False)
    for _ in range(3):
        step(g, y)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (C.c_ulonglong * 64)()
    fn = lib.hmp_debug_ktime_agg
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    fn.restype = C.c_int
    assert fn(buf) == 0
    chunks = (n_obj + 63) // 64
    per_block = (chunks + 255) // 256
    if os.environ.get("HMP_AGG_W4", "0") == "1":  # stamps of agg_fwd_w4_kernel (make EXPERIMENTS=1 KTIME=1)
        names = {20: "requests ahead + counts + far table + barrier", 24: "product (4x4x4 MFMAs) + barrier", 27: "rows (far / other edges, epilogue, stores)",
                 25: "staged registers -> LDS, counts cleared", 28: "end barrier", 29: "loop total"}
        for i, nm in names.items():
            print(f"{nm:48s} {buf[i] / 100.0:10.1f} us   ({buf[i] / 100.0 / per_block:7.2f} us per chunk, {per_block} chunks per workgroup)")
        return
    names = {20: "phase A (issue requests)", 21: "phase B+C (rows)", 22: "phase D (regs -> LDS)", 23: "barrier", 24: "loop total"}
    names.update({25: "row: root + bias", 26: "row: .. extents + lane map", 27: "row: .. loads issued", 28: "row: .. loads landed",
                  29: "row: .. sums done", 30: "row: .. stored"})
    for i, nm in names.items():
        print(f"{nm:28s} {buf[i] / 100.0:10.1f} us   ({buf[i] / 100.0 / per_block:7.2f} us per chunk, {per_block} chunks per workgroup)")


if __name__ == "__main__":
    main()
