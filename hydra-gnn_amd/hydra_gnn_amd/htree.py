"""H-tree (Neural-Tree) construction: scene graph ``HeteroData`` -> H-tree ``HeteroData``.

Host side of ``hmp_htree_*`` (csrc/htree.cpp) -- the native restatement of ``generate_htree`` + ``add_virtual_nodes_to_htree`` +
``nx_htree_to_torch`` (``src/hydra_gnn/neural_tree/construct.py:241-483``), which the reference runs on networkx both offline
(dataset preparation) and per frame in the server's ``convert_graph`` (``bin/room_classification_server:235-271``).
The library returns the typed topology; features are gathered here with tensor ops:

* object / room leaves copy ``x`` (``pos``, ``label`` / ``y`` when present) of the scene-graph node they stand for;
* clique nodes (``object-room`` / ``room-room``): ``x = [mean position of the clique's rooms | zeros]`` (construct.py:408-427
  reads the positions of the ``room_virtual`` predecessors, i.e. the member rooms), width = the feature width of the component
  type the clique was built from (``objects`` / ``rooms``), ``label = -1``;
* ``object_virtual`` / ``room_virtual`` = the original nodes (order of the scene graph), targets of the pool edges.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np
import torch

from . import _lib
from .data import HTREE_EDGE_TYPES, HTREE_INIT_EDGE_TYPES, HTREE_NODE_TYPES, HeteroData

OO = ("objects", "objects_to_objects", "objects")
RR = ("rooms", "rooms_to_rooms", "rooms")
RO = ("rooms", "rooms_to_objects", "objects")
OR = ("objects", "objects_to_rooms", "rooms")


def htree_topology(n_objects: int, n_rooms: int, oo: torch.Tensor, rr: torch.Tensor, ro: torch.Tensor) -> Dict[str, object]:
    """Typed node / edge arrays of the H-tree of one scene graph (numpy int32; see include/hydra_mp.h section 12)."""
    lib = _lib.load()

    def prep(e):
        e = e.detach().to("cpu", torch.int64).contiguous()
        return e, (e.data_ptr() if e.numel() else None), int(e.size(1))

    oo, p_oo, n_oo = prep(oo)
    rr, p_rr, n_rr = prep(rr)
    ro, p_ro, n_ro = prep(ro)
    h = C.c_void_p()
    _lib.check(lib.hmp_htree_build(int(n_objects), int(n_rooms), p_oo, n_oo, p_rr, n_rr, p_ro, n_ro, C.byref(h)))
    try:
        counts = (C.c_int32 * 4)()
        ne = (C.c_int64 * 10)()
        ni = (C.c_int64 * 3)()
        _lib.check(lib.hmp_htree_sizes(h, counts, ne, ni))
        obj = np.zeros(counts[0], dtype=np.int32)
        room = np.zeros(counts[1], dtype=np.int32)
        edges = [np.zeros((2, ne[k]), dtype=np.int32) for k in range(10)]
        init = [np.zeros((2, ni[k]), dtype=np.int32) for k in range(3)]
        pe = (C.c_void_p * 10)(*[e.ctypes.data if e.size else None for e in edges])
        pi = (C.c_void_p * 3)(*[e.ctypes.data if e.size else None for e in init])
        _lib.check(lib.hmp_htree_fill(h, obj.ctypes.data if obj.size else None, room.ctypes.data if room.size else None, pe, pi))
    finally:
        lib.hmp_htree_destroy(h)
    return {"counts": [int(c) for c in counts], "object_orig": obj, "room_orig": room, "edges": edges, "init": init}


def generate_htree(dsg: HeteroData, clique_dim: int = None) -> HeteroData:
    """``nx_htree_to_torch(add_virtual_nodes_to_htree(generate_htree(dsg)))`` of the reference for a baseline scene graph with
    node types ``objects`` / ``rooms`` (construct.py:241-483): node types ``object, room, object-room, room-room`` (+ virtual),
    the 10 message-passing edge types, pool edges ``o_to_ov`` / ``r_to_rv`` and init edges.  ``clique_dim``: keep only the
    first ``clique_dim`` feature columns of the clique nodes (the dataset drops their trailing zero block before training,
    ``mp3d_dataset.py:449-458``: 6 for MP3D)."""
    xo, xr = dsg["objects"].x, dsg["rooms"].x
    n_o, n_r = xo.size(0), xr.size(0)
    ei = dsg.edge_index_dict
    empty = torch.zeros(2, 0, dtype=torch.int64)
    ro = ei.get(RO, empty)
    if ro.size(1) == 0 and OR in ei:  # only the flipped direction is stored
        ro = ei[OR].flip(0)
    topo = htree_topology(n_o, n_r, ei.get(OO, empty), ei.get(RR, empty), ro)
    dev = xo.device
    oorig = torch.from_numpy(topo["object_orig"]).to(dev, torch.int64)
    rorig = torch.from_numpy(topo["room_orig"]).to(dev, torch.int64)
    out = HeteroData()
    out["object"].x = xo[oorig]
    out["room"].x = xr[rorig]
    pos_r = dsg["rooms"].pos if "pos" in dsg["rooms"] else xr[:, :3]

    def clique_x(n, init_rv, width):
        x = torch.zeros(n, width, dtype=xo.dtype, device=dev)
        if n and init_rv.size:
            src = torch.from_numpy(init_rv[0]).to(dev, torch.int64)
            dst = torch.from_numpy(init_rv[1]).to(dev, torch.int64)
            s = torch.zeros(n, 3, dtype=xo.dtype, device=dev).index_add_(0, dst, pos_r[src].to(xo.dtype))
            c = torch.zeros(n, dtype=xo.dtype, device=dev).index_add_(0, dst, torch.ones_like(dst, dtype=xo.dtype))
            x[:, :3] = s / c.clamp(min=1).unsqueeze(1)
        return x

    out["object-room"].x = clique_x(topo["counts"][2], topo["init"][1], clique_dim or xo.size(1))
    out["room-room"].x = clique_x(topo["counts"][3], topo["init"][2], clique_dim or xr.size(1))
    out["object_virtual"].x = xo
    out["room_virtual"].x = xr
    for t_src, t_dst, orig in (("objects", "object", oorig), ("rooms", "room", rorig)):
        for k in ("pos", "label", "y"):
            if k in dsg[t_src]:
                setattr(out[t_dst], k, getattr(dsg[t_src], k)[orig])
                setattr(out[t_dst + "_virtual"], k, getattr(dsg[t_src], k))
    for k, et in enumerate(HTREE_EDGE_TYPES):
        out[et].edge_index = torch.from_numpy(topo["edges"][k]).to(dev, torch.int64)
    for k, et in enumerate(HTREE_INIT_EDGE_TYPES):
        out[et].edge_index = torch.from_numpy(topo["init"][k]).to(dev, torch.int64)
    out["object", "o_to_ov", "object_virtual"].edge_index = torch.stack([torch.arange(oorig.numel(), device=dev), oorig])
    out["room", "r_to_rv", "room_virtual"].edge_index = torch.stack([torch.arange(rorig.numel(), device=dev), rorig])
    assert list(HTREE_NODE_TYPES) == ["object", "room", "object-room", "room-room"]
    return out
