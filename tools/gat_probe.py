#!/usr/bin/env python3
"""Kernel times of the GAT unit entry points on the two row populations of config 3 (run under rocprofv3 --kernel-trace --stats):
   A) objects <- objects: 5490 rows, in-degree ~4 + self loop       B) rooms <- objects: 447 rows, in-degree ~12 (max ~25)
H = 4, C = 128.  Each shape is launched 20 times; read the per-kernel averages off the stats CSV (shape A first, then B)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from hydra_gnn_amd import _lib  # noqa: E402
from test_gpu_ops import build_plan  # noqa: E402

DEV = "cuda:0"
lib = _lib.require_device()
rng = np.random.default_rng(0)
H, C = 4, 128
which = sys.argv[1] if len(sys.argv) > 1 else "AB"


def run(n_src, n_dst, ei, loops):
    E = ei.shape[1]
    p = build_plan(torch.from_numpy(ei).to(DEV), n_src, n_dst)
    h = torch.randn(n_src, H * C, device=DEV)
    a_s, a_d = torch.randn(n_src, 8, device=DEV), torch.randn(n_dst, 8, device=DEV)
    smax, sden = torch.zeros(n_dst, 8, device=DEV), torch.zeros(n_dst, 8, device=DEV)
    out = torch.zeros(n_dst, H * C, device=DEV)
    g = torch.randn(n_dst, H * C, device=DEV)
    n_loop = min(n_src, n_dst) if loops else 0
    al, dl = torch.zeros(E + n_loop + 1, 8, device=DEV), torch.zeros(E + n_loop + 1, 8, device=DEV)
    g_h, g_as, g_ad = torch.zeros(n_src, H * C, device=DEV), torch.zeros(n_src, 8, device=DEV), torch.zeros(n_dst, 8, device=DEV)
    args = _lib.GatArgs(H, C, int(loops), 0, 0.0, 0, 0, 0)
    for _ in range(20):
        _lib.check(lib.hmp_gat_fwd(h.data_ptr(), H * C, a_s.data_ptr(), 8, a_d.data_ptr(), 8, None, None, p["plan"], args, smax.data_ptr(),
                                   sden.data_ptr(), out.data_ptr(), H * C, _lib.stream_ptr()))
        _lib.check(lib.hmp_gat_bwd(g.data_ptr(), H * C, h.data_ptr(), H * C, a_s.data_ptr(), 8, a_d.data_ptr(), 8, None, None, p["plan"],
                                   args, smax.data_ptr(), sden.data_ptr(), al.data_ptr(), dl.data_ptr(), None, g_h.data_ptr(), H * C,
                                   g_as.data_ptr(), 8, g_ad.data_ptr(), 8, _lib.stream_ptr()))
    torch.cuda.synchronize()


if "A" in which:
    n = 5490
    E = 23000
    ei = np.stack([rng.integers(0, n, E), rng.integers(0, n, E)]).astype(np.int64)
    run(n, n, ei, True)
if "B" in which:
    n_src, n_dst = 5490, 447
    ei = np.stack([np.arange(n_src), np.sort(rng.integers(0, n_dst, n_src))]).astype(np.int64)
    print("max in-degree", np.bincount(ei[1]).max())
    run(n_src, n_dst, ei, False)
