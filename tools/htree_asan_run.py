#!/usr/bin/env python3
"""Drives libhtree_asan.so (csrc/htree.cpp built with g++ -fsanitize=address,undefined, `make -C hydra-gnn_amd/csrc asan`) over
every scene graph of the two golden files and a batch of random loopy scene graphs.  Run with libasan preloaded
(tests/test_htree_asan.py does); numpy + ctypes only, no torch, no GPU.  Prints `ASAN-OK <n graphs>` when every call returned and
the sanitizers stayed silent (they abort the process otherwise)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "hydra-gnn_amd", "hydra_gnn_amd", "libhtree_asan.so")
GOLD = os.path.join(ROOT, "tests", "golden")


def build(lib, n_obj, n_rooms, oo, rr, ro):
    arrs = [np.ascontiguousarray(a, dtype=np.int64).reshape(2, -1) for a in (oo, rr, ro)]
    h = C.c_void_p()
    args = []
    for a in arrs:
        args += [a.ctypes.data if a.size else None, a.shape[1]]
    rc = lib.hmp_htree_build(int(n_obj), int(n_rooms), *args, C.byref(h))
    if rc != 0:
        return rc, None
    counts, ne, ni = (C.c_int32 * 4)(), (C.c_int64 * 10)(), (C.c_int64 * 3)()
    assert lib.hmp_htree_sizes(h, counts, ne, ni) == 0
    obj, room = np.zeros(counts[0], np.int32), np.zeros(counts[1], np.int32)
    edges = [np.zeros((2, ne[k]), np.int32) for k in range(10)]
    init = [np.zeros((2, ni[k]), np.int32) for k in range(3)]
    pe = (C.c_void_p * 10)(*[e.ctypes.data if e.size else None for e in edges])
    pi = (C.c_void_p * 3)(*[e.ctypes.data if e.size else None for e in init])
    assert lib.hmp_htree_fill(h, obj.ctypes.data if obj.size else None, room.ctypes.data if room.size else None, pe, pi) == 0
    lib.hmp_htree_destroy(h)
    return 0, list(counts)


def main():
    lib = C.CDLL(LIB)
    VP, I32, I64 = C.c_void_p, C.c_int32, C.c_int64
    lib.hmp_htree_build.argtypes = [I32, I32, VP, I64, VP, I64, VP, I64, C.POINTER(VP)]
    lib.hmp_htree_sizes.argtypes = [VP, C.POINTER(I32), C.POINTER(I64), C.POINTER(I64)]
    lib.hmp_htree_fill.argtypes = [VP, VP, VP, C.POINTER(VP), C.POINTER(VP)]
    lib.hmp_htree_destroy.argtypes = [VP]
    lib.hmp_htree_destroy.restype = None
    n = 0
    z = np.load(os.path.join(GOLD, "htree_reference_cases.npz"))
    for name in sorted({k[:-2] for k in z.files if k.endswith("_n")}):
        no, nr = [int(v) for v in z[f"{name}_n"]]
        rc, counts = build(lib, no, nr, z[f"{name}_oo"], z[f"{name}_rr"], z[f"{name}_ro"])
        assert rc == 0 and counts == z[f"{name}_counts"].tolist(), (name, rc, counts)
        n += 1
    rng = np.random.Generator(np.random.PCG64(11))
    for it in range(200):  # random loopy scene graphs, incl. duplicate / self / one-directional edges and empty types
        nr = int(rng.integers(1, 7))
        no = int(rng.integers(0, 40))
        ro = np.stack([rng.integers(0, nr, size=no), np.arange(no)]) if no else np.zeros((2, 0))
        m = int(rng.integers(0, 3 * no + 1)) if no > 1 else 0
        oo = rng.integers(0, max(no, 1), size=(2, m))
        room_of = ro[0] if no else np.zeros(0, np.int64)
        if m:
            oo = oo[:, room_of[oo[0]] == room_of[oo[1]]]  # object edges stay inside a room (get_room_object_dsg)
        mr = int(rng.integers(0, 2 * nr + 1)) if nr > 1 else 0
        rr = rng.integers(0, nr, size=(2, mr))
        rc, counts = build(lib, no, nr, oo, rr, ro)
        assert rc == 0, (it, rc, lib.hmp_last_error())
        n += 1
    # refused inputs take the error path (no leak of the partially built object)
    bad = np.array([[0], [99]], dtype=np.int64)
    assert build(lib, 3, 1, bad, np.zeros((2, 0)), np.array([[0, 0, 0], [0, 1, 2]]))[0] != 0
    print(f"ASAN-OK {n}")


if __name__ == "__main__":
    sys.exit(main())
