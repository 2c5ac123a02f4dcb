#!/usr/bin/env python3
"""In-kernel phase times of the config-2 training step (profiling build: make -C hydra-gnn_amd/csrc KTIME=1).

    HMP_LIB=hydra-gnn_amd/hydra_gnn_amd/libhydra_mp_kt.so python tools/ktime.py

Prints, per instrumented kernel, the wall-clock deltas (us) between the KT(i) stamps of block 0 / thread 0 of the LAST
launch of that kernel in the step (see the KT(...) calls in csrc/*.hip for what each slot brackets).
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from hydra_gnn_amd import _lib, workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402


def read(lib, tag):
    buf = (C.c_ulonglong * 64)()
    fn = getattr(lib, f"hmp_debug_ktime_{tag}")
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    fn.restype = C.c_int
    assert fn(buf) == 0
    return list(buf)


def show(name, v, slots):
    t = [v[i] for i in slots]
    base = t[0]
    print(f"{name:28s}", " ".join(f"[{i}]+{(x - base) / 100.0:6.2f}" for i, x in zip(slots, t)))


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = HeterogeneousNetwork(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64,
                               num_layers=3, dropout=0.25).to(dev)
    net.train()
    batch = workloads.config2_batch(32).to(dev)
    y = batch["rooms"].y
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, use_graph=False)
    for _ in range(20):
        step(batch, y)
    torch.cuda.synchronize()
    lib = _lib.load()
    a = read(lib, "agg")
    show("agg_proj_fwd l1 (blk0)", a, [0, 1, 2])
    show("agg_row (last fwd launch)", a, [3, 4, 5, 6, 7])
    show("agg_bwd_dx l1 (blk0)", a, [8, 9, 10])
    show("front gemm tile (blk0)", read(lib, "front"), [0, 2, 3, 5, 6, 7, 13, 14, 15, 4, 1])
    show("front plan part 0 (job 0)", read(lib, "front"), [0, 8, 9, 10, 11, 12])
    show("gemm_tn_direct dW (blk0)", read(lib, "gemmd"), [8, 9])
    # the layer-0 projection alone: x[2831, 306] * Wp[192, 306]^T
    x = torch.randn(2831, 306, device=dev)
    w = torch.randn(192, 308, device=dev)[:, :306]
    z = torch.empty(2831, 192, device=dev)
    for _ in range(3):
        _lib.check(lib.hmp_gemm_f32(x.data_ptr(), 306, 0, w.data_ptr(), 308, 1, z.data_ptr(), 192, 2831, 192, 306, _lib.stream_ptr()))
    torch.cuda.synchronize()
    show("gemm layer-0 proj (blk0)", read(lib, "gemm"), [0, 1, 2, 3, 8, 9])


if __name__ == "__main__":
    main()
