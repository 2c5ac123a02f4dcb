"""Seeded synthetic scene-graph workloads for the BASELINE.json configs.

Real MP3D trajectory graphs and the word2vec table are not available offline, so the bench and the
parity tests use MP3D-*shaped* graphs: same node/edge types (reference
``src/hydra_gnn/mp3d_dataset.py:21-26``), same feature contract (objects 306-d =
pos(3) | bbox size(3) | semantic(300); rooms 6-d; 26 room labels with 25 = ignored,
``src/hydra_gnn/base_training_job.py:63``), statistics taken from the one real fixture graph
(5 rooms / 62 objects, mean object in-degree 5.74).

Everything is generated with ``numpy.random.Generator(PCG64(seed))`` so the same seed gives the
same graphs here and on the GPU box.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch

from .data import EDGE_TYPES, Data, HeteroData, collate, compute_relative_pos

BASE_SEED = 20250225
NUM_ROOM_LABELS = 26
IGNORED_LABEL = 25


def _sym_edges(pairs: np.ndarray) -> np.ndarray:
    """undirected pairs [M,2] -> directed edge_index [2, 2M] with both directions (u->v then v->u)."""
    if pairs.size == 0:
        return np.zeros((2, 0), dtype=np.int64)
    return np.concatenate([pairs.T, pairs[:, ::-1].T], axis=1).astype(np.int64)


def mp3d_like_graph(rng: np.random.Generator, sem_dim: int = 300, mean_in_degree: float = 6.0) -> HeteroData:
    n_rooms = int(rng.integers(2, 13))
    per_room = np.maximum(rng.poisson(12.0, size=n_rooms), 1)
    n_obj = int(per_room.sum())
    room_of = np.repeat(np.arange(n_rooms), per_room)
    obj_start = np.concatenate([[0], np.cumsum(per_room)])

    # objects <-> objects: within-room undirected pairs, expected in-degree ~ mean_in_degree
    pairs = []
    for r in range(n_rooms):
        k = int(per_room[r])
        if k < 2:
            continue
        n_pairs_all = k * (k - 1) // 2
        want = min(n_pairs_all, int(round(mean_in_degree * k / 2.0)))
        iu, ju = np.triu_indices(k, 1)
        sel = rng.choice(n_pairs_all, size=want, replace=False)
        sel.sort()
        pairs.append(np.stack([iu[sel], ju[sel]], 1) + obj_start[r])
    oo = _sym_edges(np.concatenate(pairs, 0) if pairs else np.zeros((0, 2), dtype=np.int64))

    # rooms <-> rooms: random spanning tree + 20 % extra undirected edges
    tree = np.array([[int(rng.integers(0, v)), v] for v in range(1, n_rooms)], dtype=np.int64).reshape(-1, 2)
    extra_n = int(round(0.2 * max(n_rooms - 1, 0)))
    have = {(int(a), int(b)) for a, b in tree}
    extra = []
    tries = 0
    while len(extra) < extra_n and tries < 50:
        tries += 1
        a, b = sorted(int(v) for v in rng.choice(n_rooms, size=2, replace=False))
        if (a, b) not in have:
            have.add((a, b))
            extra.append([a, b])
    rr_pairs = np.concatenate([tree, np.array(extra, dtype=np.int64).reshape(-1, 2)], 0)
    rr = _sym_edges(rr_pairs)

    # rooms -> objects: one edge per object; objects -> rooms is the flipped copy
    # (reference src/hydra_gnn/mp3d_dataset.py:247-249)
    ro = np.stack([room_of, np.arange(n_obj)], 0).astype(np.int64)

    def feats(n, with_sem):
        pos = rng.normal(0.0, 5.0, size=(n, 3))
        size = rng.uniform(0.1, 2.0, size=(n, 3))
        cols = [pos, size]
        if with_sem and sem_dim > 0:
            cols.append(rng.normal(0.0, 0.15, size=(n, sem_dim)))
        return np.concatenate(cols, 1).astype(np.float32), pos.astype(np.float32)

    xo, po = feats(n_obj, True)
    xr, pr = feats(n_rooms, False)

    g = HeteroData()
    g["objects"].x = torch.from_numpy(xo)
    g["objects"].pos = torch.from_numpy(po)
    g["objects"].y = torch.from_numpy(rng.integers(0, 28, size=n_obj).astype(np.int64))
    g["rooms"].x = torch.from_numpy(xr)
    g["rooms"].pos = torch.from_numpy(pr)
    g["rooms"].y = torch.from_numpy(rng.integers(0, NUM_ROOM_LABELS, size=n_rooms).astype(np.int64))
    g["objects", "objects_to_objects", "objects"].edge_index = torch.from_numpy(oo)
    g["rooms", "rooms_to_rooms", "rooms"].edge_index = torch.from_numpy(rr)
    g["objects", "objects_to_rooms", "rooms"].edge_index = torch.from_numpy(ro[::-1].copy())
    g["rooms", "rooms_to_objects", "objects"].edge_index = torch.from_numpy(ro)
    return g


def mp3d_like_batch(batch_size: int, seed: int, relative_pos: bool = False, sem_dim: int = 300) -> HeteroData:
    """Config 2 (``seed = BASE_SEED + 2``, B=32) and config 3 (``BASE_SEED + 3``, B=64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    graphs = [mp3d_like_graph(rng, sem_dim=sem_dim) for _ in range(batch_size)]
    if relative_pos:
        for g in graphs:
            compute_relative_pos(g)
    return collate(graphs)


def config2_batch(batch_size: int = 32, rank: int = 0) -> HeteroData:
    return mp3d_like_batch(batch_size, BASE_SEED + 2 + 1000 * rank)


def config3_batch(batch_size: int = 64, rank: int = 0, edge: bool = False) -> HeteroData:
    return mp3d_like_batch(batch_size, BASE_SEED + 3 + 1000 * rank, relative_pos=edge)


def big_hetero_graph(n_obj: int = 1_000_000, n_rooms: int = 10_000, deg: int = 16, feat_dim: int = 256,
                     seed: int = BASE_SEED + 5, dtype=torch.float32) -> HeteroData:
    """Config 5: one large hetero graph (object in-neighbours 90 % own room / 10 % global)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    room_of = np.sort(rng.integers(0, n_rooms, size=n_obj)).astype(np.int64)
    start = np.searchsorted(room_of, np.arange(n_rooms), side="left")
    count = np.diff(np.concatenate([start, [n_obj]]))
    dst = np.repeat(np.arange(n_obj, dtype=np.int64), deg)
    local = rng.random(n_obj * deg) < 0.9
    r = room_of[dst]
    src_local = start[r] + (rng.random(n_obj * deg) * np.maximum(count[r], 1)).astype(np.int64)
    src_glob = rng.integers(0, n_obj, size=n_obj * deg)
    src = np.where(local, np.minimum(src_local, n_obj - 1), src_glob).astype(np.int64)
    oo = np.stack([src, dst], 0)
    ro = np.stack([room_of, np.arange(n_obj, dtype=np.int64)], 0)
    n_rr = 3 * n_rooms
    a = rng.integers(0, n_rooms, size=n_rr)
    b = rng.integers(0, n_rooms, size=n_rr)
    rr = _sym_edges(np.stack([a, b], 1))
    g = HeteroData()
    g["objects"].x = torch.from_numpy(rng.normal(0, 0.5, size=(n_obj, feat_dim)).astype(np.float32)).to(dtype)
    g["rooms"].x = torch.from_numpy(rng.normal(0, 0.5, size=(n_rooms, feat_dim)).astype(np.float32)).to(dtype)
    g["rooms"].y = torch.from_numpy(rng.integers(0, NUM_ROOM_LABELS, size=n_rooms).astype(np.int64))
    g["objects", "objects_to_objects", "objects"].edge_index = torch.from_numpy(oo)
    g["rooms", "rooms_to_rooms", "rooms"].edge_index = torch.from_numpy(rr)
    g["objects", "objects_to_rooms", "rooms"].edge_index = torch.from_numpy(ro[::-1].copy())
    g["rooms", "rooms_to_objects", "objects"].edge_index = torch.from_numpy(ro)
    g.num_graphs = 1
    return g


def stanford_like_graph(rng: np.random.Generator, n_nodes: Optional[int] = None) -> Data:
    """Config 1 input: a Stanford3DSG-shaped single-room graph (room is node 0, ``x`` is
    ``[N, 6]`` float32 = pos | size, N in 2..27, every object -> room in ONE direction, object <-> object symmetric,
    15 room / 35 object classes).  The reference's ``data/Stanford3DSG.pkl`` itself is not used: both
    ``torch.load(weights_only=True)`` and ``numpy.load(allow_pickle=False)`` refuse it."""
    n = int(n_nodes if n_nodes is not None else rng.integers(2, 28))
    x = np.concatenate([rng.normal(0, 3.0, size=(n, 3)), rng.uniform(0.1, 3.0, size=(n, 3))], 1).astype(np.float32)
    k = n - 1
    obj_room = np.stack([np.arange(1, n), np.zeros(k, dtype=np.int64)], 0)
    pairs = []
    if k >= 2:
        iu, ju = np.triu_indices(k, 1)
        want = min(len(iu), int(round(1.5 * k)))
        sel = np.sort(rng.choice(len(iu), size=want, replace=False))
        pairs = np.stack([iu[sel] + 1, ju[sel] + 1], 1)
    oo = _sym_edges(np.asarray(pairs, dtype=np.int64).reshape(-1, 2))
    y = np.concatenate([rng.integers(0, 15, size=1), rng.integers(0, 35, size=k)]).astype(np.int64)
    mask = np.zeros(n, dtype=bool)
    mask[0] = True
    return Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(np.concatenate([obj_room, oo], 1).astype(np.int64)),
                y=torch.from_numpy(y), room_mask=torch.from_numpy(mask))


# ---------------------------------------------------------------------------------------------------------------------
# config 4: H-tree (Neural-Tree) batches from the committed topology fixture (tests/golden/htree_topologies.npz, made by
# tests/golden/make_htree_fixture.py with the reference's junction-tree code)
# ---------------------------------------------------------------------------------------------------------------------
from .data import HTREE_EDGE_TYPES, HTREE_INIT_EDGE_TYPES, HTREE_NODE_TYPES  # noqa: E402

HTREE_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden",
                             "htree_topologies.npz")


def htree_graph(npz, gi: int, rng: np.random.Generator) -> HeteroData:
    """One H-tree HeteroData (layout SURVEY Appendix B.2: object 306-d, room / object-room / room-room 6-d, ``room_virtual``
    carries ``num_nodes`` and ``y``; 10 message-passing edge types + the ``r_to_rv`` pool edges)."""
    counts = npz[f"g{gi}_counts"]
    n_rooms = int(npz[f"g{gi}_n_rooms"])
    g = HeteroData()
    dims = {"object": 306, "room": 6, "object-room": 6, "room-room": 6}
    for t, c in zip(HTREE_NODE_TYPES, counts):
        pos = rng.normal(0.0, 5.0, size=(int(c), 3))
        rest = rng.uniform(0.1, 2.0, size=(int(c), 3)) if t in ("object", "room") else np.zeros((int(c), 3))
        cols = [pos, rest]
        if dims[t] > 6:
            cols.append(rng.normal(0.0, 0.15, size=(int(c), dims[t] - 6)))
        g[t].x = torch.from_numpy(np.concatenate(cols, 1).astype(np.float32))
    for k, et in enumerate(HTREE_EDGE_TYPES):
        g[et].edge_index = torch.from_numpy(npz[f"g{gi}_e{k}"].astype(np.int64).reshape(2, -1))
    room_orig = npz[f"g{gi}_room_orig"].astype(np.int64)
    obj_orig = npz[f"g{gi}_object_orig"].astype(np.int64)
    n_objects = int(npz[f"g{gi}_n_objects"])
    g["room", "r_to_rv", "room_virtual"].edge_index = torch.from_numpy(np.stack([np.arange(len(room_orig)), room_orig], 0))
    g["object", "o_to_ov", "object_virtual"].edge_index = torch.from_numpy(np.stack([np.arange(len(obj_orig)), obj_orig], 0))
    # virtual nodes = copies of the original scene-graph nodes (construct.py:313-346): they carry x (read by pre_mp) and y
    g["room_virtual"].x = torch.from_numpy(np.concatenate([rng.normal(0.0, 5.0, size=(n_rooms, 3)),
                                                            rng.uniform(0.1, 2.0, size=(n_rooms, 3))], 1).astype(np.float32))
    g["object_virtual"].x = torch.from_numpy(np.concatenate([rng.normal(0.0, 5.0, size=(n_objects, 3)),
                                                              rng.uniform(0.1, 2.0, size=(n_objects, 3)),
                                                              rng.normal(0.0, 0.15, size=(n_objects, 300))], 1).astype(np.float32))
    g["room_virtual"].y = torch.from_numpy(rng.integers(0, NUM_ROOM_LABELS, size=n_rooms).astype(np.int64))
    for k, et in enumerate(HTREE_INIT_EDGE_TYPES):
        key = f"g{gi}_i{k}"
        if key in npz:
            g[et].edge_index = torch.from_numpy(npz[key].astype(np.int64).reshape(2, -1))
    return g


def htree_batch(batch_size: int = 128, seed: int = BASE_SEED + 4, fixture: Optional[str] = None) -> HeteroData:
    """Config 4 (``seed = BASE_SEED + 4``): the fixture's topologies tiled to ``batch_size`` graphs, fresh features/labels."""
    npz = np.load(fixture or HTREE_FIXTURE)
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(npz["n_graphs"])
    return collate([htree_graph(npz, i % n, rng) for i in range(batch_size)])


# ---------------------------------------------------------------------------------------------------------------------
# config 2, second input: the reference's real scene graph replicated x B
# ---------------------------------------------------------------------------------------------------------------------
DSG_FIXTURE = os.path.join(os.path.dirname(HTREE_FIXTURE), "dsg_x8F5xyUWy9e.json")


def real_fixture_frame(fixture: Optional[str] = None, seed: int = BASE_SEED) -> HeteroData:
    """The reference's test scene graph (``tests/test_data/x8F5xyUWy9e_0_gt_partial_dsg_1447.json``, static layers committed as
    ``tests/golden/dsg_x8F5xyUWy9e.json``) through the engine's own reader with the training thresholds
    (``prepare_hydra_mp3d_training.py:24-26``): 5 rooms, 62 objects, 356 + 2 + 62 + 62 directed edges.  The 300-d semantic block
    is a seeded per-label table (word2vec is a download); room labels are drawn (the bench needs a loss, not these labels).
    Object connectivity runs on the device (``csrc/dsg.hip``): needs the GPU."""
    from . import dsg

    rog = dsg.RoomObjectGraph(dsg.load_dsg_json(fixture or DSG_FIXTURE))
    rng = np.random.Generator(np.random.PCG64(seed))
    table = rng.normal(0.0, 0.15, size=(int(rog.sg.label[rog.objects].max()) + 1, 300))
    oo = dsg.object_connectivity(rog, 1.5, 2.0, 0.2)
    frame = dsg.to_hetero_data(rog, oo, {"objects": table[rog.sg.label[rog.objects]]}).to("cpu")
    frame["rooms"].y = torch.from_numpy(rng.integers(0, NUM_ROOM_LABELS, size=len(rog.rooms)))
    return frame


def real_fixture_batch(batch_size: int = 32, fixture: Optional[str] = None) -> HeteroData:
    return collate([real_fixture_frame(fixture)] * batch_size)
