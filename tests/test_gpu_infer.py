"""Inference hot loop: ``model.predict(data)`` = the reference server's
``self.model(data.to(device)).argmax(dim=1).cpu()`` (bin/room_classification_server:285-286) through the native eval forward,
``hmp_argmax_rows`` and one pinned D2H of the labels.  Labels are index work: bit-exact against ``forward().argmax()``, and the
logits they come from carry the 1e-5 oracle parity of test_gpu_models / test_gpu_gat / test_gpu_htree."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import ops, workloads  # noqa: E402
from hydra_gnn_amd.data import collate, collate_homogeneous  # noqa: E402
from hydra_gnn_amd.models import (HeterogeneousNetwork, HeterogeneousNeuralTreeNetwork, HomogeneousNetwork)  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("n,c", [(1, 1), (5, 26), (300, 15), (4097, 26), (64, 130)])
def test_argmax_rows_first_maximum(n, c):
    g = torch.Generator().manual_seed(n * 31 + c)
    x = torch.randint(-3, 4, (n, c), generator=g).float()  # many ties
    want = x.argmax(dim=1)
    # torch documents no tie rule for argmax; pin ours (first maximum) explicitly
    first = torch.stack([(row == row.max()).nonzero()[0, 0] for row in x])
    got = ops.argmax_rows(x.to(DEV)).cpu()
    assert torch.equal(got, first)
    assert torch.equal(x.gather(1, got[:, None]), x.gather(1, want[:, None]))
    pad = torch.full((n, c + 3), 100.0)  # a leading dimension larger than the row: the padding never wins
    pad[:, :c] = x
    assert torch.equal(ops.argmax_rows(pad.to(DEV)[:, :c]).cpu(), first)


def hetero(block):
    torch.manual_seed(1)
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, dropout=0.25)
    if block == "GraphSAGE":
        kw.update(conv_block="GraphSAGE", hidden_dim=64, num_layers=3)
    else:
        kw.update(conv_block="GAT", GAT_hidden_dims=[32, 32], GAT_heads=[2, 2, 2], GAT_concats=[True, True, False])
    return HeterogeneousNetwork(**kw).to(DEV).eval()


@pytest.mark.parametrize("block", ["GraphSAGE", "GAT"])
@pytest.mark.parametrize("n_graphs", [1, 7])
def test_predict_equals_forward_argmax(block, n_graphs):
    net = hetero(block)
    for seed in (3, 4, 5):  # consecutive frames of different sizes through one workspace
        batch = workloads.mp3d_like_batch(n_graphs, seed).to(DEV)
        with torch.no_grad():
            want = net(batch).argmax(dim=1).cpu()
        got = net.predict(batch)
        assert got.dtype == torch.int64 and not got.is_cuda
        assert torch.equal(got, want)


def test_predict_htree_and_homogeneous():
    torch.manual_seed(2)
    dims = {"object": 306, "room": 6, "object-room": 6, "room-room": 6, "object_virtual": 306, "room_virtual": 6}
    tree = HeterogeneousNeuralTreeNetwork(input_dim_dict=dims, output_dim=26,
                                          conv_block="GraphSAGE", hidden_dim=32, num_layers=3, disable_initialization=True).to(DEV).eval()
    batch = workloads.htree_batch(5, seed=9).to(DEV)
    with torch.no_grad():
        want = tree(batch).argmax(dim=1).cpu()
    assert torch.equal(tree.predict(batch), want) and want.numel() == int(batch["room_virtual"].num_nodes)
    rng = np.random.Generator(np.random.PCG64(7))
    graphs = collate_homogeneous([workloads.stanford_like_graph(rng) for _ in range(6)]).to(DEV)
    for block in ("GraphSAGE", "GCN", "GIN"):
        net = HomogeneousNetwork(input_dim=6, output_dim=15, conv_block=block, hidden_dim=32, num_layers=3).to(DEV).eval()
        with torch.no_grad():
            want = net(graphs).argmax(dim=1).cpu()
        assert torch.equal(net.predict(graphs), want) and want.numel() == 6


def test_predict_leaves_training_state_alone():
    """a predict() between forward and backward would overwrite the activations: the engine must notice"""
    from hydra_gnn_amd import _lib

    net = hetero("GraphSAGE").train()
    batch = workloads.mp3d_like_batch(3, 3).to(DEV)
    pred = net(batch)
    net.predict(batch)
    with pytest.raises(_lib.HydraMPError):
        pred.sum().backward()


def test_predict_reuses_the_plan_across_frames_with_unchanged_topology():
    """Same edge tensors, new features (a frame whose objects moved but whose graph did not change): predict() skips the plan
    build (hmp_batch.plan_valid) and returns what a full rebuild returns; any change of the edge tensors rebuilds."""
    import copy as _copy

    torch.manual_seed(3)
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3, dropout=0.25)
    net = HeterogeneousNetwork(**kw).to(DEV).eval()
    ref = HeterogeneousNetwork(**kw).to(DEV).eval()
    ref.load_state_dict(net.state_dict())
    frame = collate([workloads.mp3d_like_graph(np.random.default_rng(8))]).to(DEV)
    a0 = net.predict(frame).clone()
    assert net.native()._plan_key is not None
    for k in range(3):
        frame["objects"].x = frame["objects"].x + 0.05 * torch.randn_like(frame["objects"].x)  # new features, same edge tensors
        fresh = _copy.copy(frame)
        got = net.predict(frame).clone()  # plan reused
        want = ref.predict(collate([frame.to("cpu")]).to(DEV)).clone()  # new tensors: plan rebuilt
        assert torch.equal(got, want), k
    # an in-place edit of an edge list bumps its version: the plan is rebuilt and the answer follows
    et = ("objects", "objects_to_objects", "objects")
    ei = frame[et].edge_index
    ei[:, 0] = ei[:, 1]
    got = net.predict(frame).clone()
    want = ref.predict(collate([frame.to("cpu")]).to(DEV)).clone()
    assert torch.equal(got, want)
    # a training step in between invalidates the reuse
    step = net.train_step(lr=0.001, ignored_label=25, use_graph=False)
    net.train(); step(frame, frame["rooms"].y); net.eval()
    assert net.native()._plan_key is None
    ref.load_state_dict(net.state_dict())
    assert torch.equal(net.predict(frame), ref.predict(collate([frame.to("cpu")]).to(DEV)))


def test_predict_rebuilds_the_plan_when_a_new_frame_lands_on_recycled_addresses():
    """the server builds fresh edge tensors per frame; once frame k is freed the caching allocator hands frame k+1 the
    same addresses (same shape, `_version` 0).  Reuse is decided by tensor IDENTITY (the previous frame's edge tensors are kept
    alive by the net), so the changed connectivity is seen: the answer equals a full rebuild on a fresh net."""
    torch.manual_seed(5)
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3, dropout=0.25)
    net = HeterogeneousNetwork(**kw).to(DEV).eval()
    ref = HeterogeneousNetwork(**kw).to(DEV).eval()
    ref.load_state_dict(net.state_dict())
    host = collate([workloads.mp3d_like_graph(np.random.default_rng(21))])
    et_or = ("objects", "objects_to_rooms", "rooms")
    et_ro = ("rooms", "rooms_to_objects", "objects")
    n_rooms = int(host["rooms"].x.size(0))
    assert n_rooms >= 2
    frame = host.to(DEV)
    ptrs = {e: frame[e].edge_index.data_ptr() for e in frame.edge_types}
    net.predict(frame)
    del frame
    # frame k+1: every object re-assigned to the next room (same counts, same shapes, different connectivity)
    host[et_or].edge_index = torch.stack([host[et_or].edge_index[0], (host[et_or].edge_index[1] + 1) % n_rooms])
    host[et_ro].edge_index = host[et_or].edge_index.flip([0])
    frame2 = host.to(DEV)
    recycled = sum(frame2[e].edge_index.data_ptr() == ptrs[e] for e in frame2.edge_types)
    got = net.predict(frame2).clone()
    want = ref.predict(host.to(DEV)).clone()
    assert torch.equal(got, want), f"stale plan used ({recycled} edge tensors on recycled addresses)"
