"""CPU tests of the input contract: the reference's own known answers for ``fill_missing_edge_index``
(``tests/test_mp3d_dataset.py:143-180`` -- the only tensor-level vectors the reference's tests hold for this path),
PyG ``Batch.from_data_list`` collation offsets, ``compute_relative_pos``
(``src/hydra_gnn/mp3d_dataset.py:298-319``) and the committed H-tree topology fixture."""
import os

import numpy as np
import torch

from hydra_gnn_amd import workloads
from hydra_gnn_amd.data import (EDGE_TYPES, HTREE_EDGE_TYPES, HeteroData, collate, compute_relative_pos,
                                fill_missing_edge_index)


def test_fill_missing_edge_index_reference_case_1_room():
    # reference tests/test_mp3d_dataset.py:145-161
    data = HeteroData()
    data["rooms"].x = torch.rand((1, 6))
    data["objects"].x = torch.rand((3, 306))
    object_edge = torch.tensor([[0, 1, 1, 2, 2, 0], [1, 0, 2, 1, 0, 2]])
    room_object_edge = torch.tensor([[0, 0, 0], [0, 1, 2]])
    data["objects", "objects_to_objects", "objects"].edge_index = object_edge
    data["rooms", "rooms_to_objects", "objects"].edge_index = room_object_edge
    fill_missing_edge_index(data, edge_types=EDGE_TYPES)
    assert ("rooms", "rooms_to_rooms", "rooms") in data.edge_index_dict
    assert data[("rooms", "rooms_to_rooms", "rooms")].num_edges == 0
    assert data[("rooms", "rooms_to_rooms", "rooms")].edge_index.dtype == torch.int64
    assert ("objects", "objects_to_rooms", "rooms") in data.edge_index_dict
    assert torch.all(data[("objects", "objects_to_rooms", "rooms")].edge_index == room_object_edge.flip([0]))


def test_fill_missing_edge_index_reference_case_2_rooms():
    # reference tests/test_mp3d_dataset.py:163-180
    data = HeteroData()
    data["rooms"].x = torch.rand((2, 6))
    data["objects"].x = torch.rand((4, 306))
    room_edge = torch.tensor([[0, 1], [1, 0]])
    object_edge = torch.tensor([[0, 1, 1, 2, 2, 0], [1, 0, 2, 1, 0, 2]])
    room_object_edge = torch.tensor([[0, 0, 0, 1], [0, 1, 2, 3]])
    data["rooms", "rooms_to_rooms", "rooms"].edge_index = room_edge
    data["objects", "objects_to_objects", "objects"].edge_index = object_edge
    data["rooms", "rooms_to_objects", "objects"].edge_index = room_object_edge
    fill_missing_edge_index(data, edge_types=EDGE_TYPES)
    assert data[("rooms", "rooms_to_rooms", "rooms")].num_edges == 2
    assert torch.all(data[("objects", "objects_to_rooms", "rooms")].edge_index == room_object_edge.flip([0]))


def test_fill_missing_inter_type_without_reverse_is_empty():
    data = HeteroData()
    data["rooms"].x = torch.rand((1, 6))
    data["objects"].x = torch.rand((2, 306))
    fill_missing_edge_index(data, edge_types=EDGE_TYPES)
    for et in EDGE_TYPES:
        assert data[et].edge_index.shape == (2, 0)


def test_collate_offsets_follow_source_and_destination_types():
    g1 = workloads.mp3d_like_graph(np.random.default_rng(0))
    g2 = workloads.mp3d_like_graph(np.random.default_rng(1))
    b = collate([g1, g2])
    no1, nr1 = g1["objects"].num_nodes, g1["rooms"].num_nodes
    assert b["objects"].num_nodes == no1 + g2["objects"].num_nodes
    assert torch.equal(b["objects"].x[no1:], g2["objects"].x)
    et = ("objects", "objects_to_rooms", "rooms")
    e1 = g1[et].num_edges
    assert torch.equal(b[et].edge_index[:, :e1], g1[et].edge_index)
    assert torch.equal(b[et].edge_index[0, e1:], g2[et].edge_index[0] + no1)   # source rows offset by #objects
    assert torch.equal(b[et].edge_index[1, e1:], g2[et].edge_index[1] + nr1)   # destination rows offset by #rooms
    assert torch.equal(b["rooms"].ptr, torch.tensor([0, nr1, nr1 + g2["rooms"].num_nodes]))
    assert b.num_graphs == 2
    assert torch.equal(b["rooms"].y, torch.cat([g1["rooms"].y, g2["rooms"].y]))


def test_compute_relative_pos():
    g = workloads.mp3d_like_graph(np.random.default_rng(2))
    pos_o, pos_r = g["objects"].pos.clone(), g["rooms"].pos.clone()
    x_o = g["objects"].x.clone()
    compute_relative_pos(g)
    assert g["objects"].x.shape[1] == 303 and g["rooms"].x.shape[1] == 3
    assert torch.equal(g["objects"].x, x_o[:, 3:])
    ei = g["objects", "objects_to_rooms", "rooms"].edge_index
    assert torch.equal(g["objects", "objects_to_rooms", "rooms"].edge_attr, pos_r[ei[1]] - pos_o[ei[0]])


def test_config2_workload_statistics():
    b = workloads.config2_batch(32)
    n_o, n_r = b["objects"].num_nodes, b["rooms"].num_nodes
    assert b["objects"].x.shape == (n_o, 306) and b["rooms"].x.shape == (n_r, 6)
    assert 32 * 2 <= n_r <= 32 * 12
    oo = b["objects", "objects_to_objects", "objects"].edge_index
    assert abs(oo.shape[1] / n_o - 6.0) < 1.0           # mean in-degree ~6 (fixture graph: 5.74)
    key = lambda e: set(map(tuple, e.t().tolist()))
    assert key(oo) == key(oo.flip([0]))                 # stored in both directions
    ro = b["rooms", "rooms_to_objects", "objects"].edge_index
    assert ro.shape[1] == n_o and torch.equal(torch.sort(ro[1]).values, torch.arange(n_o))
    assert torch.equal(b["objects", "objects_to_rooms", "rooms"].edge_index, ro.flip([0]))
    assert int(b["rooms"].y.max()) <= 25
    # same seed -> same batch (what the GPU box regenerates)
    b2 = workloads.config2_batch(32)
    assert torch.equal(b2["objects"].x, b["objects"].x) and torch.equal(b2["rooms"].y, b["rooms"].y)


def test_htree_fixture_is_consistent():
    assert os.path.exists(workloads.HTREE_FIXTURE)
    npz = np.load(workloads.HTREE_FIXTURE)
    n = int(npz["n_graphs"])
    assert n >= 4
    for gi in range(n):
        counts = npz[f"g{gi}_counts"]
        src_t = {"object": 0, "room": 1, "object-room": 2, "room-room": 3}
        for k, (s, _, t) in enumerate(HTREE_EDGE_TYPES):
            e = npz[f"g{gi}_e{k}"].reshape(2, -1)
            if e.shape[1]:
                assert e[0].max() < counts[src_t[s]] and e[1].max() < counts[src_t[t]] and e.min() >= 0
        # every room leaf is a copy of an original room; every original room has at least one leaf
        ro = npz[f"g{gi}_room_orig"]
        assert len(ro) == counts[1] and set(ro.tolist()) == set(range(int(npz[f"g{gi}_n_rooms"])))
        oo = npz[f"g{gi}_object_orig"]
        assert set(oo.tolist()) == set(range(int(npz[f"g{gi}_n_objects"])))
        # leaf <-> clique relations are stored in both directions
        a = npz[f"g{gi}_e0"].reshape(2, -1)
        b = npz[f"g{gi}_e1"].reshape(2, -1)
        assert set(map(tuple, a.T.tolist())) == set(map(tuple, b[::-1].T.tolist()))
    batch = workloads.htree_batch(4, seed=1)
    assert batch["room_virtual"].num_nodes == batch["room_virtual"].y.numel()
    pool = batch["room", "r_to_rv", "room_virtual"].edge_index
    assert int(pool[1].max()) < batch["room_virtual"].num_nodes and pool.shape[1] == batch["room"].num_nodes


def test_hetero_htree_to_homogeneous_follows_the_reference_layout():
    """heterogeneous_htree_to_homogeneous (reference mp3d_dataset.py:73-119): node types concatenated in store order with
    zero-padded features, init edges = edges leaving a virtual node, pool edges = edges entering one, the rest (the H-tree
    itself) stays in edge_index and is undirected; room_mask / object_mask mark the virtual nodes; y = -1 elsewhere."""
    from hydra_gnn_amd.data import heterogeneous_htree_to_homogeneous

    npz = np.load(workloads.HTREE_FIXTURE)
    g = workloads.htree_graph(npz, 0, np.random.Generator(np.random.PCG64(3)))
    counts = {t: g[t].num_nodes for t in g.node_types}
    n_init = sum(g[e].num_edges for e in g.edge_types if e[0] in ("object_virtual", "room_virtual"))
    n_pool = sum(g[e].num_edges for e in g.edge_types if e[2] in ("object_virtual", "room_virtual"))
    n_tree = sum(g[e].num_edges for e in HTREE_EDGE_TYPES)
    y_rv = g["room_virtual"].y.clone()
    d = heterogeneous_htree_to_homogeneous(g)
    assert d.x.shape == (sum(counts.values()), 306)
    assert d.init_edge_index.shape[1] == n_init and d.pool_edge_index.shape[1] == n_pool and d.edge_index.shape[1] == n_tree
    assert int(d.room_mask.sum()) == counts["room_virtual"] and int(d.object_mask.sum()) == counts["object_virtual"]
    assert torch.equal(d.y[d.room_mask], y_rv) and bool((d.y[~d.room_mask] == -1).all())
    # 6-d node types are zero padded to 306
    order = [t for t in g.node_types]
    off = np.cumsum([0] + [counts[t] for t in order])
    i_room = order.index("room")
    assert bool((d.x[off[i_room]:off[i_room + 1], 6:] == 0).all())
    # the H-tree part is undirected: every edge has its reverse
    fwd = set(map(tuple, d.edge_index.t().tolist()))
    assert all((b, a) in fwd for a, b in fwd)
    # init edges start at virtual nodes, pool edges end there
    virt = d.room_mask | d.object_mask
    assert bool(virt[d.init_edge_index[0]].all()) and bool(virt[d.pool_edge_index[1]].all())


def test_oracle_homogeneous_htree_network_matches_its_definition():
    """oracle HomogeneousNeuralTreeNetwork = pre_mp GAT on init edges (every node) -> convs on edge_index -> LeafPool mean over
    pool edges -> rows of room_mask (reference homogeneous_neural_tree_network.py:75-103), checked against a hand-rolled
    composition of the oracle operators"""
    from hydra_gnn_amd.data import collate_homogeneous, heterogeneous_htree_to_homogeneous
    from oracle import models as omodels
    from oracle.pyg_ref import LeafPool

    npz = np.load(workloads.HTREE_FIXTURE)
    rng = np.random.Generator(np.random.PCG64(4))
    gs = []
    for i in range(2):
        d = heterogeneous_htree_to_homogeneous(workloads.htree_graph(npz, i, rng))
        del d.__dict__["edge_type"]
        d.x = d.x[:, :6].contiguous()
        gs.append(d)
    b = collate_homogeneous(gs)
    torch.manual_seed(0)
    m = omodels.HomogeneousNeuralTreeNetwork(6, output_dim=26, conv_block="GraphSAGE", hidden_dim=8, num_layers=2, dropout=0.0).eval()
    x = m.pre_mp(b.x, b.init_edge_index)
    x = torch.relu(m.convs[0](x, b.edge_index))
    x = m.convs[1](x, b.edge_index)
    x = LeafPool()(x, b.pool_edge_index)
    torch.testing.assert_close(m(b), x[b.room_mask])
    assert set(m.state_dict()) >= {"pre_mp.att_src", "pre_mp.att_dst", "pre_mp.lin_src.weight", "pre_mp.bias", "convs.0.lin_l.weight"}


def test_collate_records_where_each_graph_starts():
    """collate() keeps [PyG] Batch.ptr per node type and, per edge type, the offset of every graph's first edge (what PyG holds in
    Batch._slice_dict): edges of graph g are entries [ptr[g], ptr[g+1]) and join nodes of graph g only -- the statement the plan
    build of a small batch relies on (hmp_batch::d_node_ptr / d_edge_ptr).  A graph may lack nodes or edges of a type."""
    gs = [workloads.mp3d_like_graph(np.random.Generator(np.random.PCG64(s))) for s in range(5)]
    empty = HeteroData()  # two objects, no rooms, no edges; the same attribute set as the others
    for t, n in (("objects", 2), ("rooms", 0)):
        for k in gs[0][t].keys():
            v = getattr(gs[0][t], k)
            if isinstance(v, torch.Tensor):
                setattr(empty[t], k, torch.zeros((n,) + tuple(v.shape[1:]), dtype=v.dtype))
    for et in EDGE_TYPES:
        empty[et].edge_index = torch.zeros(2, 0, dtype=torch.int64)
    gs.insert(2, empty)
    b = collate(gs)
    for t in b.node_types:
        counts = [g[t].num_nodes for g in gs]
        assert b[t].ptr.tolist() == np.concatenate([[0], np.cumsum(counts)]).tolist()
    for et in b.edge_types:
        counts = [g[et].edge_index.size(1) for g in gs]
        ptr = b[et].ptr
        assert ptr.dtype == torch.int64 and ptr.tolist() == np.concatenate([[0], np.cumsum(counts)]).tolist()
        src, _, dst = et
        ei = b[et].edge_index
        for g in range(len(gs)):  # every edge of the slice joins rows of graph g
            sl = ei[:, int(ptr[g]):int(ptr[g + 1])]
            if sl.numel():
                assert int(sl[0].min()) >= int(b[src].ptr[g]) and int(sl[0].max()) < int(b[src].ptr[g + 1])
                assert int(sl[1].min()) >= int(b[dst].ptr[g]) and int(sl[1].max()) < int(b[dst].ptr[g + 1])
    assert b.max_graph_nodes == max(max(g[t].num_nodes for t in g.node_types) for g in gs)
    del b[EDGE_TYPES[0]].ptr  # a caller that permutes an edge list drops the statement
    assert "ptr" not in b[EDGE_TYPES[0]]


def test_edge_offsets_follow_the_edge_index_they_were_computed_for():
    """collate() attaches per-graph edge offsets (`ptr`) that let the plan build read slices; they describe ONE edge_index.
    Replacing edge_index drops them; an in-place edit leaves a version stamp that no longer matches (the engine then describes the
    batch without `ptr`), and `.to()` does not carry a stale `ptr` over."""
    from hydra_gnn_amd import workloads
    from hydra_gnn_amd.data import collate

    rng = np.random.Generator(np.random.PCG64(3))
    batch = collate([workloads.mp3d_like_graph(rng) for _ in range(3)])
    et = ("objects", "objects_to_objects", "objects")
    assert "ptr" in batch[et] and batch[et].ptr_version == batch[et].edge_index._version
    moved = batch.to("cpu")
    assert "ptr" in moved[et] and moved[et].ptr_version == moved[et].edge_index._version
    # in-place edit: the stamp no longer matches; a copy made now carries no offsets
    batch[et].edge_index[0, 0] = batch[et].edge_index[0, 1]
    assert batch[et].ptr_version != batch[et].edge_index._version
    assert "ptr" not in batch.to("cpu")[et]
    # replacement: offsets dropped at once
    batch[et].edge_index = batch[et].edge_index[:, :-2].contiguous()
    assert "ptr" not in batch[et] and "ptr_version" not in batch[et]
