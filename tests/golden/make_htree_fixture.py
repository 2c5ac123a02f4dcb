#!/usr/bin/env python3
"""Generates tests/golden/htree_topologies.npz -- H-tree (Neural-Tree) topologies for config-4 inputs.

Run in the BUILD container only (needs /root/reference and networkx):

    python tests/golden/make_htree_fixture.py

The junction-tree hierarchies come from the REFERENCE's own importable code,
``hydra_gnn.neural_tree.generate_junction_tree_hierarchies.sample_and_generate_jth`` (pure networkx).
The surrounding assembly (``src/hydra_gnn/neural_tree/construct.py:87-236,241-371``: per-room object trees, room copies
on the lowest cliques, root hookup, virtual nodes / pool edges, node-type based edge typing :450-468) cannot be
imported (it needs torch_geometric), so it is restated here procedure for procedure.  The output is DATA only
(typed node lists + typed directed edge lists of a handful of seeded MP3D-like scene graphs).
"""
import os
import sys

import networkx as nx
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "hydra-gnn_amd"), "/root/reference/src"]

from hydra_gnn.neural_tree.generate_junction_tree_hierarchies import sample_and_generate_jth  # noqa: E402  (reference)
from hydra_gnn_amd import workloads  # noqa: E402

NODE_TYPES = ["object", "room", "object-room", "room-room"]
EDGE_TYPES = [("object", "object-room"), ("object-room", "object"), ("room", "object-room"), ("object-room", "room"),
              ("room", "room-room"), ("room-room", "room"), ("object-room", "room-room"), ("room-room", "object-room"),
              ("object-room", "object-room"), ("room-room", "room-room")]


def scene_to_nx(g):
    """undirected networkx graph of one MP3D-like HeteroData: objects then rooms, attribute node_type / orig."""
    n_obj, n_room = g["objects"].num_nodes, g["rooms"].num_nodes
    G = nx.Graph()
    for i in range(n_obj):
        G.add_node(i, node_type="object", orig=i, x=[0.0], pos=[0.0] * 3, label=0)
    for r in range(n_room):
        G.add_node(n_obj + r, node_type="room", orig=r, x=[0.0], pos=[0.0] * 3, label=0)
    for s, d in g["objects", "objects_to_objects", "objects"].edge_index.t().tolist():
        G.add_edge(s, d)
    for s, d in g["rooms", "rooms_to_rooms", "rooms"].edge_index.t().tolist():
        G.add_edge(n_obj + s, n_obj + d)
    for s, d in g["rooms", "rooms_to_objects", "objects"].edge_index.t().tolist():
        G.add_edge(n_obj + s, d)
    return G


def component_jth(Gc, kind, room=None):
    """construct.py:87-188 (generate_component_jth)."""
    _, jth, roots = sample_and_generate_jth(Gc, k=1000, zero_feature=[0.0], copy_node_attributes=["x", "pos", "label", "node_type"],
                                            need_root_tree=True, remove_edges_every_layer=True)
    if kind == "rooms":
        if len(Gc) == 1:
            roots = [0]
        for _, d in jth.nodes.items():
            if d["type"] == "clique":
                d["node_type"] = "room-room"
        return jth, roots
    r, rdata = room
    if len(Gc) == 1:
        jth.add_node(1, x=[0.0], pos=[0.0] * 3, type="clique", clique_has=[jth.nodes[0]["clique_has"]])
        jth.add_edge(0, 1)
        roots = [1]
    for _, d in jth.nodes.items():
        if d["type"] == "clique":
            d["node_type"] = "object-room"
            d["clique_has"].append(r)
    leaves = [i for i, d in jth.nodes.items() if d["type"] == "node"]
    lowest = set(sum([[n for n in jth.neighbors(leaf)] for leaf in leaves], []))
    idx = jth.number_of_nodes()
    for c in lowest:
        jth.add_node(idx, x=rdata["x"], pos=rdata["pos"], label=rdata["label"], node_type=rdata["node_type"], orig=rdata["orig"],
                     type="node", clique_has=r)
        jth.add_edge(c, idx)
        idx += 1
    return jth, roots


def build_htree(G):
    """construct.py:241-310 (generate_htree) incl. the HTree helper :191-236."""
    out = nx.Graph()
    for comp in nx.connected_components(G):
        Gc = G.subgraph(comp).copy()
        rooms = [i for i in Gc.nodes if Gc.nodes[i]["node_type"] == "room"]
        jth, room_roots = component_jth(Gc.subgraph(rooms).copy() if False else nx.Graph(Gc.subgraph(rooms)), "rooms")
        n_nodes = jth.number_of_nodes()
        for r in rooms:
            objs = [i for i in Gc.neighbors(r) if Gc.nodes[i]["node_type"] == "object"]
            Go = Gc.subgraph(objs)
            for oc in nx.connected_components(Go):
                ojth, oroots = component_jth(nx.Graph(G.subgraph(oc)), "objects", room=(r, Gc.nodes[r]))
                for e in list(ojth.subgraph(oroots).edges):
                    ojth.remove_edge(*e)
                for root in room_roots:
                    ch = jth.nodes[root]["clique_has"]
                    if not isinstance(ch, list):
                        jth.nodes[root]["clique_has"] = ch = [ch]
                    if r in ch:
                        relabel = dict(zip(range(ojth.number_of_nodes()), range(n_nodes, n_nodes + ojth.number_of_nodes())))
                        g2 = nx.relabel_nodes(ojth.copy(), relabel)
                        jth = nx.compose(jth, g2)
                        n_nodes = jth.number_of_nodes()
                        for orr in oroots:
                            jth.add_edge(root, relabel[orr])
        if len(rooms) == 1 and jth.number_of_nodes() > 1:
            assert jth.nodes[0]["type"] == "node" and len(jth.nodes[0]["clique_has"]) == 1
            jth.nodes[0]["clique_has"] = jth.nodes[0]["clique_has"][0]
        out = nx.disjoint_union(out, jth)
    return out


def typed_arrays(G, ht):
    """node-type ids, original index of every leaf, typed directed edges, pool edges (construct.py:313-371,450-468)."""
    n_obj = sum(1 for _, d in G.nodes.items() if d["node_type"] == "object")
    per_type = {t: [] for t in NODE_TYPES}
    local = {}
    for i in range(ht.number_of_nodes()):
        t = ht.nodes[i]["node_type"]
        local[i] = len(per_type[t])
        per_type[t].append(i)
    leaf_orig = {}
    for t in ("object", "room"):
        o = []
        for i in per_type[t]:
            ch = ht.nodes[i]["clique_has"]
            assert not isinstance(ch, list)
            o.append(ch if t == "object" else ch - n_obj)
        leaf_orig[t] = np.array(o, dtype=np.int32)
    edges = {et: [] for et in EDGE_TYPES}
    for u, v in ht.to_directed().edges:
        et = (ht.nodes[u]["node_type"], ht.nodes[v]["node_type"])
        edges[et].append((local[u], local[v]))
    # init edges virtual -> clique (construct.py:364-369): one per member of the clique; the virtual index is the member's
    # original index inside its type (virtual nodes are ordered objects then rooms, construct.py:338-346)
    init = {"ov_to_or": [], "rv_to_or": [], "rv_to_rr": []}
    for t, key_o, key_r in (("object-room", "ov_to_or", "rv_to_or"), ("room-room", None, "rv_to_rr")):
        for i in per_type[t]:
            for member in ht.nodes[i]["clique_has"]:
                if member < n_obj:
                    assert key_o is not None
                    init[key_o].append((member, local[i]))
                else:
                    init[key_r].append((member - n_obj, local[i]))
    return per_type, leaf_orig, edges, init


def main():
    rng = np.random.Generator(np.random.PCG64(workloads.BASE_SEED + 4))
    out = {}
    n_graphs = 6
    for gi in range(n_graphs):
        # sparse object graphs (mean in-degree 2): random graphs of the full MP3D density (~6) have a large treewidth
        # and blow the hierarchy up 15x; at degree 2 the H-tree has ~7x the nodes of the scene graph
        g = workloads.mp3d_like_graph(rng, sem_dim=0, mean_in_degree=2.0)
        G = scene_to_nx(g)
        ht = build_htree(G)
        # NB not necessarily a forest: networkx's junction_tree shares one sepset node between all cliques that
        # intersect in the same set, and the reference projects that bipartite graph onto the cliques
        per_type, leaf_orig, edges, init = typed_arrays(G, ht)
        for k, name in enumerate(("ov_to_or", "rv_to_or", "rv_to_rr")):
            out[f"g{gi}_i{k}"] = np.array(init[name], dtype=np.int32).reshape(-1, 2).T
        out[f"g{gi}_n_objects"] = np.int32(g["objects"].num_nodes)
        out[f"g{gi}_n_rooms"] = np.int32(g["rooms"].num_nodes)
        out[f"g{gi}_counts"] = np.array([len(per_type[t]) for t in NODE_TYPES], dtype=np.int32)
        out[f"g{gi}_object_orig"] = leaf_orig["object"]
        out[f"g{gi}_room_orig"] = leaf_orig["room"]
        for k, et in enumerate(EDGE_TYPES):
            e = np.array(edges[et], dtype=np.int32).reshape(-1, 2).T
            out[f"g{gi}_e{k}"] = e
        print(f"graph {gi}: {g['objects'].num_nodes} objects, {g['rooms'].num_nodes} rooms -> htree nodes {out[f'g{gi}_counts'].tolist()}, "
              f"edges {sum(len(v) for v in edges.values())}")
    out["n_graphs"] = np.int32(n_graphs)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "htree_topologies.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
