"""Device-side collation: GraphStore.collate == data.collate(...).to(device), bit for bit, for the MP3D
hetero layout, the H-tree layout (count-only node types, 15 edge types) and GAT_edge batches (edge_attr); the collated batch
drives the native step.  Byte / index work: the bar is bit-exact."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import workloads  # noqa: E402
from hydra_gnn_amd.data import collate, compute_relative_pos  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402
from hydra_gnn_amd.store import GraphStore  # noqa: E402

DEV = "cuda:0"


def assert_same(a, b):
    assert a.node_types == b.node_types and a.edge_types == b.edge_types and a.num_graphs == b.num_graphs
    for t in a.node_types:
        keys = sorted(k for k in b[t].keys())
        assert sorted(k for k in a[t].keys()) == keys, (t, sorted(a[t].keys()), keys)
        for k in keys:
            va, vb = getattr(a[t], k), getattr(b[t], k)
            if isinstance(vb, torch.Tensor):
                assert va.dtype == vb.dtype and va.shape == vb.shape, (t, k)
                assert torch.equal(va.cpu(), vb.cpu()), (t, k)
            else:
                assert va == vb, (t, k)
    for e in a.edge_types:
        assert torch.equal(a[e].edge_index.cpu(), b[e].edge_index.cpu()), e
        assert ("edge_attr" in a[e]) == ("edge_attr" in b[e])
        if "edge_attr" in b[e]:
            assert torch.equal(a[e].edge_attr.cpu(), b[e].edge_attr.cpu()), e


def mp3d_graphs(n, seed=5, rel_pos=False):
    rng = np.random.default_rng(seed)
    gs = [workloads.mp3d_like_graph(rng) for _ in range(n)]
    if rel_pos:
        for g in gs:
            compute_relative_pos(g)
    return gs


@pytest.mark.parametrize("ids", [[0], [3, 1, 4, 1, 5], list(range(12)), [11, 0]])
def test_store_collate_is_bit_identical_to_host_collate(ids):
    gs = mp3d_graphs(12)
    store = GraphStore(gs, DEV)
    assert_same(store.collate(ids), collate([gs[i] for i in ids]))


def test_store_collate_with_edge_attr_and_htree_layout():
    gs = mp3d_graphs(6, seed=9, rel_pos=True)
    store = GraphStore(gs, DEV)
    assert_same(store.collate([5, 2, 2, 0]), collate([gs[i] for i in (5, 2, 2, 0)]))
    npz = np.load(workloads.HTREE_FIXTURE)
    rng = np.random.Generator(np.random.PCG64(8))
    hs = [workloads.htree_graph(npz, i % int(npz["n_graphs"]), rng) for i in range(7)]
    hstore = GraphStore(hs, DEV)
    assert_same(hstore.collate([6, 0, 3]), collate([hs[i] for i in (6, 0, 3)]))


def test_collated_batches_drive_the_native_step_and_beat_host_collation():
    """a fresh random batch every step from the resident store; also reports the per-batch cost next to host collate + H2D"""
    gs = mp3d_graphs(96, seed=21)
    store = GraphStore(gs, DEV)
    torch.manual_seed(0)
    net = HeterogeneousNetwork({"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3,
                               dropout=0.25).to(DEV)
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=1)
    rng = np.random.default_rng(0)
    losses = []
    for _ in range(40):
        b = store.collate(rng.choice(96, size=32, replace=False))
        step(b, b["rooms"].y)
        losses.append(step.loss())
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])
    _, status = net.native().read_state()
    assert status == 0
    ids = [rng.choice(96, size=32, replace=False) for _ in range(30)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in ids:
        store.collate(i)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / len(ids)
    t0 = time.perf_counter()
    for i in ids:
        collate([gs[j] for j in i]).to(DEV)
    torch.cuda.synchronize()
    t_host = (time.perf_counter() - t0) / len(ids)
    print(f"collate per 32-graph batch: device store {1e3 * t_dev:.3f} ms, host collate + H2D {1e3 * t_host:.3f} ms")
    assert t_dev < t_host


@pytest.mark.parametrize("batch_size,n_graphs,rel_pos", [(4, 12, False), (7, 9, True), (96, 120, False)])
def test_batch_stream_one_launch_collation_is_bit_identical(batch_size, n_graphs, rel_pos):
    """store.BatchStream (hmp_collator_run: one host call, one kernel, tables in the kernel's argument block for small batches
    and through the pinned ring for large ones) == host collate, bit for bit, batch after batch in the same buffers."""
    gs = mp3d_graphs(n_graphs, seed=21, rel_pos=rel_pos)
    store = GraphStore(gs, DEV)
    kw = dict(input_dim_dict={"objects": 303 if rel_pos else 306, "rooms": 3 if rel_pos else 6}, output_dim=26,
              conv_block="GAT_edge" if rel_pos else "GraphSAGE", hidden_dim=16, num_layers=3, GAT_hidden_dims=[8, 8], GAT_heads=[2, 2, 2],
              GAT_concats=[True, True, False], dropout=0.0)
    torch.manual_seed(0)
    net = HeterogeneousNetwork(**kw).to(DEV)
    stream = store.stream(net, batch_size, "rooms")
    rng = np.random.default_rng(1)
    for it in range(12):  # > the ring depth of 8
        ids = rng.choice(n_graphs, size=batch_size if it % 3 else max(batch_size // 2, 1), replace=True).tolist()
        h = stream.next(ids)
        got = stream.data()
        ref = collate([gs[i] for i in ids])
        for t in ref.node_types:
            for k in ref[t].keys():
                vb = getattr(ref[t], k)
                if isinstance(vb, torch.Tensor) and k not in ("batch", "ptr"):
                    assert torch.equal(getattr(got[t], k).cpu(), vb), (it, t, k)
        for e in ref.edge_types:
            assert torch.equal(got[e].edge_index.cpu(), ref[e].edge_index), (it, e)
            if "edge_attr" in ref[e]:
                assert torch.equal(got[e].edge_attr.cpu(), ref[e].edge_attr), (it, e)
        assert int(h.c.n_out) == ref["rooms"].x.size(0)
        # the offset tables the stream hands to the plan build (hmp_batch::d_node_ptr / d_edge_ptr) are collate's ptr vectors
        B = len(ids)
        off = stream._offsets.view(-1, stream._off_stride).cpu()
        for t in ref.node_types:
            assert torch.equal(off[stream._slot_of[t], :B + 1], ref[t].ptr), (it, t)
        for e in ref.edge_types:
            assert torch.equal(off[stream._slot_of[e], :B + 1], ref[e].ptr), (it, e)
        assert int(h.c.n_graphs) == B


def test_batch_stream_drives_the_native_step_like_host_collated_batches():
    """TrainStep.run(stream.next(ids)) == TrainStep(collate(...).to(device)) step for step: same losses, same parameters."""
    gs = mp3d_graphs(40, seed=4)
    store = GraphStore(gs, DEV)
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3, dropout=0.25)
    torch.manual_seed(0)
    net_a = HeterogeneousNetwork(**kw).to(DEV)
    torch.manual_seed(0)
    net_b = HeterogeneousNetwork(**kw).to(DEV)
    net_a.train(); net_b.train()
    step_a = net_a.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=3, use_graph=False)
    step_b = net_b.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=3, use_graph=False)
    stream = store.stream(net_a, 8, "rooms")
    rng = np.random.default_rng(2)
    for it in range(10):
        ids = rng.choice(40, size=8, replace=False).tolist()
        step_a.run(stream.next(ids))
        b = collate([gs[i] for i in ids]).to(DEV)
        step_b(b, b["rooms"].y)
        assert step_a.loss() == step_b.loss(), it
    for (n, p), (_, q) in zip(net_a.named_parameters(), net_b.named_parameters()):
        assert torch.equal(p, q), n
    assert step_a.steps_taken() == 10 and net_a.native().read_state()[1] == 0
