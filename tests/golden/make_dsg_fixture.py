#!/usr/bin/env python3
"""Generates tests/golden/dsg_x8F5xyUWy9e.json (inputs) and tests/golden/dsg_x8F5xyUWy9e_expected.npz (expected outputs) for
the scene-graph reader of ``hydra_gnn_amd/dsg.py``.

Run in the BUILD container only (needs /root/reference):

    python tests/golden/make_dsg_fixture.py

Inputs: the one real scene graph the reference's tests hold, ``tests/test_data/x8F5xyUWy9e_0_gt_partial_dsg_1447.json`` (a
spark_dsg JSON dump: 65 objects, 423 places, 5 rooms, 1 building, 1445 agent poses, mesh).  It is DATA; the copy written here
keeps what the room-object conversion reads (static nodes of layers 2-5 with id / layer / position / bounding box / semantic
label / name, and the edges among them) and drops the agent trajectory, the mesh and the per-place mesh bookkeeping
(2.2 MB -> ~0.3 MB).

Expected outputs: produced by the REFERENCE's own functions ``hydra_gnn.preprocess_dsgs.get_room_object_dsg`` and
``add_object_connectivity`` (``src/hydra_gnn/preprocess_dsgs.py:191-292``), imported from /root/reference and driven with a
duck-typed stand-in for the ``spark_dsg`` bindings (the C++ package is not installable offline).  The stand-in implements only
the container behaviour those two functions touch, following spark_dsg's documented semantics: layers iterate in ascending
node id, ``get_parent`` is the neighbour in the layer above, ``siblings`` the same-layer neighbours in ascending id,
``insert_edge`` refuses duplicates.  The geometric predicates and every decision are the reference's code.
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC_JSON = "/root/reference/tests/test_data/x8F5xyUWy9e_0_gt_partial_dsg_1447.json"
OUT_JSON = os.path.join(ROOT, "tests", "golden", "dsg_x8F5xyUWy9e.json")
OUT_NPZ = os.path.join(ROOT, "tests", "golden", "dsg_x8F5xyUWy9e_expected.npz")
STATIC_TYPES = {"ObjectNodeAttributes", "PlaceNodeAttributes", "RoomNodeAttributes", "SemanticNodeAttributes"}


# ---- duck-typed spark_dsg ---------------------------------------------------------------------------------------------------
class _Id:
    def __init__(self, v):
        self.value = v

    def __repr__(self):
        return str(self.value)


class _BBox:
    def __init__(self, lo, hi):
        self.min, self.max = np.array(lo, dtype=np.float64), np.array(hi, dtype=np.float64)


class _Attrs:
    def __init__(self, a):
        self.position = np.array(a["position"], dtype=np.float64)
        self.bounding_box = _BBox(a["bounding_box"]["min"], a["bounding_box"]["max"])
        self.semantic_label = a["semantic_label"]
        self.name = a.get("name", "")


class _Node:
    def __init__(self, graph, nid, layer, attrs):
        self._g, self.id, self.layer, self.attributes = graph, _Id(nid), layer, attrs

    def get_parent(self):
        ps = sorted(n for n in self._g.adj[self.id.value] if self._g.nodes[n].layer > self.layer)
        return ps[0] if ps else None

    def has_parent(self):
        return self.get_parent() is not None

    def siblings(self):
        return sorted(n for n in self._g.adj[self.id.value] if self._g.nodes[n].layer == self.layer)


class _Layer:
    def __init__(self, graph, layer):
        self._g, self._l = graph, layer

    @property
    def nodes(self):
        return [self._g.nodes[k] for k in sorted(self._g.nodes) if self._g.nodes[k].layer == self._l]


class DynamicSceneGraph:
    def __init__(self):
        self.nodes, self.adj, self.edge_log = {}, {}, []

    def add_node(self, layer, nid, attrs):
        if nid in self.nodes:
            return False
        self.nodes[nid] = _Node(self, nid, layer, attrs)
        self.adj[nid] = set()
        return True

    def insert_edge(self, a, b):
        if a not in self.nodes or b not in self.nodes or a == b or b in self.adj[a]:
            return False
        self.adj[a].add(b)
        self.adj[b].add(a)
        self.edge_log.append((a, b))
        return True

    def get_layer(self, layer):
        return _Layer(self, layer)

    def get_node(self, nid):
        return self.nodes[nid]

    def get_position(self, nid):
        return self.nodes[nid].attributes.position


shim = types.ModuleType("spark_dsg")
shim.DynamicSceneGraph = DynamicSceneGraph
shim.DsgLayers = types.SimpleNamespace(OBJECTS=2, PLACES=3, ROOMS=4, BUILDINGS=5)


def main():
    raw = json.load(open(SRC_JSON))
    keep = [n for n in raw["nodes"] if n["attributes"]["type"] in STATIC_TYPES]
    ids = {n["id"] for n in keep}
    edges = [(e["source"], e["target"]) for e in raw["edges"] if e["source"] in ids and e["target"] in ids]
    reduced = {
        "directed": raw["directed"], "multigraph": raw["multigraph"], "layer_ids": raw["layer_ids"],
        "nodes": [{"id": n["id"], "layer": n["layer"],
                   "attributes": {"type": n["attributes"]["type"], "name": n["attributes"].get("name", ""),
                                  "position": n["attributes"]["position"], "semantic_label": n["attributes"]["semantic_label"],
                                  "bounding_box": {"min": n["attributes"]["bounding_box"]["min"],
                                                   "max": n["attributes"]["bounding_box"]["max"]}}} for n in keep],
        "edges": [{"source": s, "target": t} for s, t in edges],
    }
    with open(OUT_JSON, "w") as f:
        json.dump(reduced, f, separators=(",", ":"))

    # ---- the reference's functions on the stand-in container
    sys.path.insert(0, "/root/reference/src")
    import hydra_gnn.preprocess_dsgs as ref  # noqa: E402  (reference; its torch_geometric import is guarded)

    ref.get_spark_dsg = lambda return_mp3d=False: shim
    G = DynamicSceneGraph()
    for n in reduced["nodes"]:
        G.add_node(n["layer"], n["id"], _Attrs(n["attributes"]))
    for e in reduced["edges"]:
        G.insert_edge(e["source"], e["target"])
    import time

    t0 = time.perf_counter()
    G_ro = ref.get_room_object_dsg(G, verbose=False)
    n_before = len(G_ro.edge_log)
    t1 = time.perf_counter()
    # the thresholds of the inference server (bin/room_classification_server:213-215,253-257)
    ref.add_object_connectivity(G_ro, threshold_near=1.5, max_near=2.0, max_on=0.2)
    t2 = time.perf_counter()
    print(f"reference on this host: get_room_object_dsg {1e3 * (t1 - t0):.1f} ms, add_object_connectivity {1e3 * (t2 - t1):.1f} ms")
    oo = G_ro.edge_log[n_before:]

    objs = G_ro.get_layer(2).nodes
    rooms = G_ro.get_layer(4).nodes
    oid = {n.id.value: i for i, n in enumerate(objs)}
    rid = {n.id.value: i for i, n in enumerate(rooms)}
    all_obj = [n["id"] for n in reduced["nodes"] if n["layer"] == 2]
    rr = [(a, b) for a, b in G_ro.edge_log[:n_before] if a in rid and b in rid]
    np.savez_compressed(
        OUT_NPZ,
        obj_id=np.array([n.id.value for n in objs], dtype=np.uint64),
        obj_pos=np.stack([n.attributes.position for n in objs]),
        obj_bb_min=np.stack([n.attributes.bounding_box.min for n in objs]),
        obj_bb_max=np.stack([n.attributes.bounding_box.max for n in objs]),
        obj_label=np.array([n.attributes.semantic_label for n in objs], dtype=np.int64),
        obj_room=np.array([rid[n.get_parent()] for n in objs], dtype=np.int64),
        dropped_obj_id=np.array(sorted(set(all_obj) - set(oid)), dtype=np.uint64),
        room_id=np.array([n.id.value for n in rooms], dtype=np.uint64),
        room_pos=np.stack([n.attributes.position for n in rooms]),
        room_label=np.array([n.attributes.semantic_label for n in rooms], dtype=np.int64),
        rr_edges=np.array([[rid[a], rid[b]] for a, b in rr], dtype=np.int64).T.reshape(2, -1),
        oo_edges=np.array([[oid[a], oid[b]] for a, b in oo], dtype=np.int64).T.reshape(2, -1),  # (node, earlier node of its room)
        thresholds=np.array([1.5, 2.0, 0.2]),
    )
    print(f"{len(objs)} objects kept of {len(all_obj)}, {len(rooms)} rooms, {len(rr)} room-room edges, {len(oo)} object-object edges; "
          f"{os.path.getsize(OUT_JSON) / 1e3:.0f} kB json")


if __name__ == "__main__":
    main()
