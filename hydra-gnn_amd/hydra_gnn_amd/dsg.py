"""Scene-graph reader: spark_dsg JSON -> room-object graph -> the ``HeteroData`` the models read.

What the reference does per frame in ``GnnModel.convert_graph`` (``bin/room_classification_server:235-271``) and per file in
``Hydra_mp3d_data`` (``src/hydra_gnn/mp3d_dataset.py:130-284``), on the ``spark_dsg`` C++ bindings + numpy:

1. ``get_room_object_dsg`` (``preprocess_dsgs.py:228-292``): keep the rooms and their sibling edges; attach every object to the
   room of its parent place, or -- when that place has no room -- to the room of the nearest sibling place that has one; drop
   objects with neither.  Index bookkeeping over a few hundred nodes: done here on the host, in the reference's visiting order
   (ascending node id) so that ties resolve identically.
2. ``add_object_connectivity`` (``:191-225``): pairwise geometric predicates between the objects of a room, an O(n^2) Python
   loop -> ``hmp_object_edges_count`` / ``hmp_object_edges_fill`` (``csrc/dsg.hip``), float64, bit-exact edge set.
3. ``to_torch`` + ``fill_missing_edge_index`` (``mp3d_dataset.py:267-274``): ``x = [position | bbox size | semantic]``
   (``preprocess_dsgs.py:401-421``), the four ``EDGE_TYPES``; intra-layer edges in both directions, rooms_to_objects one edge
   per object and objects_to_rooms its flip.

Pinned by ``tests/golden/dsg_x8F5xyUWy9e_expected.npz`` -- the reference's own functions run on the reference's test graph
(``tests/golden/make_dsg_fixture.py``).  Not reproducible offline and therefore NOT built: the word2vec block of the object
features (the reference downloads GoogleNews vectors; pass ``semantic`` rows yourself or use the 6-d ``--remove_word2vec``
contract) and ``spark_dsg.add_bounding_boxes_to_layer`` (C++, absent): room boxes are taken as the AABB of the positions of
the room's places, stated here as an assumption.
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from .data import HeteroData

OBJECTS, PLACES, ROOMS, BUILDINGS = 2, 3, 4, 5
_STATIC = ("ObjectNodeAttributes", "PlaceNodeAttributes", "RoomNodeAttributes", "SemanticNodeAttributes")


class SceneGraph:
    """Static layers of a spark_dsg JSON dump: per node id / layer / position / bounding box / semantic label, and adjacency."""

    def __init__(self, ids, layer, pos, bb_min, bb_max, label, adj):
        self.ids, self.layer, self.pos, self.bb_min, self.bb_max, self.label, self.adj = ids, layer, pos, bb_min, bb_max, label, adj
        self.index = {int(v): i for i, v in enumerate(ids)}

    def of_layer(self, layer: int) -> np.ndarray:
        """node indices of a layer in ascending node id (the iteration order of spark_dsg's layer containers)"""
        idx = np.nonzero(self.layer == layer)[0]
        return idx[np.argsort(self.ids[idx], kind="stable")]

    def parent(self, i: int) -> int:
        """neighbour in a higher layer (a node has at most one; the lowest id wins if the dump holds several), -1: none"""
        ps = [j for j in self.adj[i] if self.layer[j] > self.layer[i]]
        return min(ps, key=lambda j: self.ids[j]) if ps else -1

    def siblings(self, i: int) -> List[int]:
        return sorted((j for j in self.adj[i] if self.layer[j] == self.layer[i]), key=lambda j: self.ids[j])


def load_dsg_json(src: Union[str, dict]) -> SceneGraph:
    """Parse a spark_dsg JSON dump (``DynamicSceneGraph.save``): static nodes only (agent poses share layer id 2 with the
    objects in the dump but live in a dynamic layer of their own), edges among them, mesh ignored."""
    raw = src if isinstance(src, dict) else json.load(open(src))
    nodes = [n for n in raw["nodes"] if n["attributes"].get("type") in _STATIC]
    ids = np.array([n["id"] for n in nodes], dtype=np.uint64)
    index = {int(v): i for i, v in enumerate(ids)}
    adj: List[set] = [set() for _ in nodes]
    for e in raw["edges"]:
        a, b = index.get(e["source"]), index.get(e["target"])
        if a is None or b is None or a == b:
            continue
        adj[a].add(b)
        adj[b].add(a)
    f = lambda key: np.array([key(n["attributes"]) for n in nodes], dtype=np.float64).reshape(len(nodes), 3)
    return SceneGraph(ids, np.array([n["layer"] for n in nodes], dtype=np.int64), f(lambda a: a["position"]),
                      f(lambda a: a["bounding_box"]["min"]), f(lambda a: a["bounding_box"]["max"]),
                      np.array([n["attributes"]["semantic_label"] for n in nodes], dtype=np.int64), adj)


class RoomObjectGraph:
    """``get_room_object_dsg`` result: rooms, kept objects (ascending id), each object's room, room-room edges."""

    def __init__(self, sg: SceneGraph):
        self.sg = sg
        rooms = sg.of_layer(ROOMS)
        room_index = {int(r): k for k, r in enumerate(rooms)}
        rr: List[Tuple[int, int]] = []
        seen = set()
        for r in rooms:  # preprocess_dsgs.py:236-246
            for s in sg.siblings(int(r)):
                key = (min(int(r), s), max(int(r), s))
                if key not in seen:
                    seen.add(key)
                    rr.append((room_index[int(r)], room_index[s]))
        kept, room_of, dropped = [], [], []
        for o in sg.of_layer(OBJECTS):  # :248-283
            place = sg.parent(int(o))
            if place < 0:
                dropped.append(int(o))
                continue
            room = sg.parent(place)
            if room < 0:
                cands = [s for s in sg.siblings(place) if sg.parent(s) >= 0]
                if not cands:
                    dropped.append(int(o))
                    continue
                dist = [float(np.linalg.norm(sg.pos[place] - sg.pos[s])) for s in cands]
                room = sg.parent(cands[int(np.argsort(dist, kind="stable")[0])])  # list.sort by distance is stable
            kept.append(int(o))
            room_of.append(room_index[room])
        self.rooms, self.objects = rooms, np.array(kept, dtype=np.int64)
        self.obj_room = np.array(room_of, dtype=np.int64)
        self.dropped = np.array(dropped, dtype=np.int64)
        self.rr_edges = np.array(rr, dtype=np.int64).reshape(-1, 2).T
        # room boxes: AABB of the positions of the room's places (assumption, see the module docstring)
        self.room_bb = np.zeros((len(rooms), 2, 3))
        for k, r in enumerate(rooms):
            kids = [j for j in sg.adj[int(r)] if sg.layer[j] == PLACES]
            if kids:
                p = sg.pos[kids]
                self.room_bb[k, 0], self.room_bb[k, 1] = p.min(0), p.max(0)

    @property
    def obj_pos(self):
        return self.sg.pos[self.objects]

    @property
    def obj_size(self):
        return self.sg.bb_max[self.objects] - self.sg.bb_min[self.objects]


def object_connectivity(rog: RoomObjectGraph, threshold_near: float = 2.0, max_near: float = 2.0, max_on: float = 0.2,
                        device="cuda:0") -> torch.Tensor:
    """``add_object_connectivity``: int64 ``[2, E]`` on the device, column = (object, earlier object of its room), objects
    indexed in ``rog.objects`` order; the reference's insertion order (by object, then by earlier object)."""
    lib = _lib.require_device()
    dev = torch.device(device)
    n = int(rog.objects.size)
    pos = torch.from_numpy(np.ascontiguousarray(rog.obj_pos)).to(dev)
    size = torch.from_numpy(np.ascontiguousarray(rog.obj_size)).to(dev)
    room = torch.from_numpy(rog.obj_room.astype(np.int32)).to(dev)
    count = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    offset = torch.empty(n + 1, dtype=torch.int32, device=dev)
    args = (pos.data_ptr(), size.data_ptr(), room.data_ptr(), n, float(threshold_near), float(max_near), float(max_on))
    with torch.cuda.device(dev):
        st = _lib.stream_ptr()
        _lib.check(lib.hmp_object_edges_count(*args, count.data_ptr(), offset.data_ptr(), st))
        total = int(offset[n].item())  # the one host round trip: the edge list is allocated to size
        edges = torch.empty((2, total), dtype=torch.int32, device=dev)
        _lib.check(lib.hmp_object_edges_fill(*args, offset.data_ptr(), edges.data_ptr() if total else None, total, st))
    return edges.to(torch.int64)


def to_hetero_data(rog: RoomObjectGraph, oo_edges: torch.Tensor, semantic: Optional[Dict[str, np.ndarray]] = None,
                   device="cuda:0", dtype=torch.float32) -> HeteroData:
    """``to_torch(use_heterogeneous=True)`` + ``fill_missing_edge_index``: x = [position | bbox size | semantic rows if given],
    ``pos``, ``label`` (Hydra semantic id), ``node_ids``; EDGE_TYPES per SURVEY Appendix B.1."""
    sg, dev = rog.sg, torch.device(device)
    g = HeteroData()

    def feats(pos, size, sem):
        cols = [pos, size] + ([sem] if sem is not None else [])
        return torch.from_numpy(np.concatenate(cols, 1)).to(dtype).to(dev)

    sem = semantic or {}
    g["objects"].x = feats(rog.obj_pos, rog.obj_size, sem.get("objects"))
    g["objects"].pos = torch.from_numpy(rog.obj_pos).to(dtype).to(dev)
    g["objects"].label = torch.from_numpy(sg.label[rog.objects]).to(dev)
    g["objects"].node_ids = torch.from_numpy(sg.ids[rog.objects].astype(np.int64)).to(dev)
    rpos = sg.pos[rog.rooms]
    g["rooms"].x = feats(rpos, rog.room_bb[:, 1] - rog.room_bb[:, 0], sem.get("rooms"))
    g["rooms"].pos = torch.from_numpy(rpos).to(dtype).to(dev)
    g["rooms"].label = torch.from_numpy(sg.label[rog.rooms]).to(dev)
    g["rooms"].node_ids = torch.from_numpy(sg.ids[rog.rooms].astype(np.int64)).to(dev)
    both = lambda e: torch.cat([e, e.flip(0)], dim=1)
    rr = torch.from_numpy(rog.rr_edges).to(dev)
    ro = torch.stack([torch.from_numpy(rog.obj_room), torch.arange(rog.objects.size)]).to(dev)
    g["objects", "objects_to_objects", "objects"].edge_index = both(oo_edges.to(dev))
    g["rooms", "rooms_to_rooms", "rooms"].edge_index = both(rr)
    g["rooms", "rooms_to_objects", "objects"].edge_index = ro
    g["objects", "objects_to_rooms", "rooms"].edge_index = ro.flip(0)
    return g


def frame_to_data(src: Union[str, dict], threshold_near: float = 1.5, max_near: float = 2.0, max_on: float = 0.2,
                  semantic: Optional[Dict[str, np.ndarray]] = None, device="cuda:0") -> Tuple[HeteroData, RoomObjectGraph]:
    """``GnnModel.convert_graph`` for the baseline hetero model with the server's thresholds
    (bin/room_classification_server:213-215): JSON frame -> device ``HeteroData`` (6-d features unless ``semantic`` rows are given)."""
    rog = RoomObjectGraph(load_dsg_json(src))
    oo = object_connectivity(rog, threshold_near, max_near, max_on, device)
    return to_hetero_data(rog, oo, semantic, device), rog
