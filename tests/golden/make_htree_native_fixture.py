#!/usr/bin/env python3
"""Generates tests/golden/htree_reference_cases.npz: H-trees built by the REFERENCE's own junction-tree-hierarchy code
(``sample_and_generate_jth``, importable) + the assembly restated in make_htree_fixture.py (construct.py needs torch_geometric),
for small scene graphs of several structural classes.  tests/test_htree_native.py compares csrc/htree.cpp against them.

Run in the BUILD container only:  python tests/golden/make_htree_native_fixture.py
"""
import os
import sys

import networkx as nx
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_htree_fixture as ref  # noqa: E402  (imports the reference's generate_junction_tree_hierarchies)


def scene(n_obj_per_room, obj_graph_fn, room_graph_fn, seed):
    """rooms 0..R-1 with their objects; returns (n_obj, n_rooms, oo pairs, rr pairs, ro pairs)"""
    rng = np.random.default_rng(seed)
    R = len(n_obj_per_room)
    oo, ro, base = [], [], 0
    for r, k in enumerate(n_obj_per_room):
        for a, b in obj_graph_fn(k, rng):
            oo.append((base + a, base + b))
        for i in range(k):
            ro.append((r, base + i))
        base += k
    rr = room_graph_fn(R, rng)
    return base, R, oo, rr, ro


def path(k, rng): return [(i, i + 1) for i in range(k - 1)]
def star(k, rng): return [(0, i) for i in range(1, k)]
def complete(k, rng): return [(i, j) for i in range(k) for j in range(i + 1, k)]
def rtree(k, rng): return [(int(rng.integers(0, i)), i) for i in range(1, k)]
def empty(k, rng): return []


def sparse(k, rng):
    e = set()
    for i in range(1, k):
        e.add((int(rng.integers(0, i)), i))
    for _ in range(max(1, k // 2)):
        a, b = sorted(rng.integers(0, k, size=2).tolist())
        if a != b:
            e.add((a, b))
    return sorted(e)


CASES = {
    "paths": ([3, 4, 2], path, path, 0), "stars": ([5, 4], star, path, 1), "cliques": ([3, 4, 5], complete, complete, 2),
    "rtrees": ([6, 7, 5, 4], rtree, rtree, 3), "singletons": ([1, 1, 1], empty, path, 4), "one_room": ([5], rtree, empty, 5),
    "isolated_objects": ([3, 2], empty, path, 6), "rtrees_big": ([12, 9, 10, 8, 11], rtree, rtree, 7),
    "loopy_a": ([8, 7, 6], sparse, sparse, 8), "loopy_b": ([10, 9, 9, 8], sparse, sparse, 9), "loopy_c": ([14, 12], sparse, path, 10),
}


def main():
    out = {"names": np.array(list(CASES))}
    for name, (sizes, og, rg, seed) in CASES.items():
        n_obj, n_rooms, oo, rr, ro = scene(sizes, og, rg, seed)
        G = nx.Graph()
        for i in range(n_obj):
            G.add_node(i, node_type="object", orig=i, x=[0.0], pos=[0.0] * 3, label=0)
        for r in range(n_rooms):
            G.add_node(n_obj + r, node_type="room", orig=r, x=[0.0], pos=[0.0] * 3, label=0)
        G.add_edges_from(oo)
        G.add_edges_from((n_obj + a, n_obj + b) for a, b in rr)
        G.add_edges_from((n_obj + a, b) for a, b in ro)
        ht = ref.build_htree(G)
        per_type, leaf_orig, edges, init = ref.typed_arrays(G, ht)
        out[f"{name}_n"] = np.array([n_obj, n_rooms], dtype=np.int32)
        out[f"{name}_oo"] = np.array(oo, dtype=np.int64).reshape(-1, 2).T
        out[f"{name}_rr"] = np.array(rr, dtype=np.int64).reshape(-1, 2).T
        out[f"{name}_ro"] = np.array(ro, dtype=np.int64).reshape(-1, 2).T
        out[f"{name}_counts"] = np.array([len(per_type[t]) for t in ref.NODE_TYPES], dtype=np.int32)
        out[f"{name}_object_orig"] = leaf_orig["object"]
        out[f"{name}_room_orig"] = leaf_orig["room"]
        for k, et in enumerate(ref.EDGE_TYPES):
            out[f"{name}_e{k}"] = np.array(edges[et], dtype=np.int32).reshape(-1, 2).T
        for k, key in enumerate(("ov_to_or", "rv_to_or", "rv_to_rr")):
            out[f"{name}_i{k}"] = np.array(init[key], dtype=np.int32).reshape(-1, 2).T
        print(name, "scene", n_obj, n_rooms, "-> htree", out[f"{name}_counts"].tolist())
    path_out = os.path.join(HERE, "htree_reference_cases.npz")
    np.savez_compressed(path_out, **out)
    print("wrote", path_out, os.path.getsize(path_out), "bytes")


if __name__ == "__main__":
    main()
