"""Aggregate-first SAGE convs (hmp_conv_spec.agg_first; csrc/aggfirst.hip): a conv whose destination type is much
smaller than its source type is evaluated in [PyG] SAGEConv's own order -- mean of the source rows, then lin_l on the destination
rows -- with the gradient of that mean entering the source rows before their activation mask.  Exact algebra: everything against
the float64 oracle at 1e-5, at sizes the oracle holds; HMP_AGG_FIRST=1 forces the choice the engine makes by itself from 32 768
source rows on (tests/test_gpu_config5.py runs it at that size, in fp32 and bf16 mode)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import _lib, workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork, HeterogeneousNeuralTreeNetwork  # noqa: E402
from oracle import models as omodels  # noqa: E402

ATOL, RTOL = 1e-5, 1e-5
DEV = "cuda:0"
OR = ("objects", "objects_to_rooms", "rooms")
RO = ("rooms", "rooms_to_objects", "objects")


def pair(kw, seed=0, cls=(omodels.HeterogeneousNetwork, HeterogeneousNetwork)):
    torch.manual_seed(seed)
    ora = cls[0](**kw)
    net = cls[1](**kw)
    net.load_state_dict(ora.state_dict(), strict=True)
    return ora, net.to(DEV)


def replay_fn(net):
    lib = _lib.require_device()

    def replay(x, pp, training, tag):
        if not training or pp == 0:
            return x
        layer, t = tag[1:].split(".", 1)
        n, F = x.shape
        m = torch.zeros(max(n * F, 1), dtype=torch.uint8, device=DEV)
        if n * F:
            _lib.check(lib.hmp_dropout_mask(net._seed, net._rng_step, net._drop_stream(int(layer), t), pp, n, F, m.data_ptr(), _lib.stream_ptr()))
        return x * m[: n * F].view(n, F).cpu().to(x.dtype) / (1.0 - pp)

    return replay


def check_against_oracle(ora, net, batch, label_type="rooms", train=True):
    net.train(train)
    pred = net(batch.to(DEV))
    ora.dropout_fn = replay_fn(net)
    o64 = copy.deepcopy(ora).double()
    o64.train(train)
    b64 = batch.to("cpu")
    for t in b64.node_types:
        if "x" in b64[t]:
            b64[t].x = b64[t].x.double()
    pred_ref = o64(b64)
    y = batch[label_type].y
    loss_ref = o64.loss(pred_ref, y, y != 25)
    loss_ref.backward()
    torch.testing.assert_close(pred.detach().cpu().double(), pred_ref.detach(), atol=ATOL, rtol=RTOL)
    yg = y.to(DEV)
    loss = net.loss(pred, yg, yg != 25)
    torch.testing.assert_close(loss.detach().cpu().double(), loss_ref.detach(), atol=ATOL, rtol=RTOL)
    loss.backward()
    og = dict(o64.named_parameters())
    n = 0
    for name, p in net.named_parameters():
        ref = og[name].grad
        if ref is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        torch.testing.assert_close(p.grad.cpu().double(), ref, atol=ATOL, rtol=RTOL, msg=lambda m: f"{name}: {m}")
        n += 1
    return pred.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}, n


@pytest.mark.parametrize("hidden,layers,dropout", [(64, 3, 0.25), (128, 2, 0.0), (256, 3, 0.25)])
def test_aggregate_first_matches_the_oracle_and_the_project_first_engine(monkeypatch, hidden, layers, dropout):
    """MP3D contract (objects 306-d at pitch 306: the unaligned input path of the segment mean; rooms 6-d), every layer's
    objects -> rooms conv aggregate-first incl. the last one (objects then have NO projection in that layer: their gradient is the
    masked transpose alone), training mode with replayed masks.  And the same net with HMP_AGG_FIRST=0 agrees at 1e-5."""
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=hidden, num_layers=layers, dropout=dropout)
    batch = workloads.mp3d_like_batch(9, seed=41)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HMP_AGG_FIRST", mode)
        ora, net = pair(kw, seed=2)
        pred, grads, n = check_against_oracle(ora, net, batch, train=dropout > 0)
        nat = net.native()
        got = {(l, tuple(nat.layers[l].convs[c].edge_type)) for (l, c), on in nat._agg_first.items() if on}
        # forced: every live conv between two node types (rooms -> objects too; dead in the last layer, whose objects are not read)
        want = {(l, OR) for l in range(layers)} | {(l, RO) for l in range(layers - 1)}
        assert got == (want if mode == "1" else set()), got
        assert nat.read_state()[1] == 0 and n >= 8
        res[mode] = (pred, grads)
    monkeypatch.delenv("HMP_AGG_FIRST")


def test_aggregate_first_fused_training_step_tracks_the_oracle(monkeypatch):
    """the native training step (plan + fwd + CE + bwd + Adam) with aggregate-first convs against oracle + torch.optim.Adam over
    several steps on changing batches (the weight-gradient slabs of the conv, the gather-add, the standalone transpose)"""
    monkeypatch.setenv("HMP_AGG_FIRST", "1")
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3, dropout=0.0)
    ora, net = pair(kw, seed=5)
    net.train()
    lr = 0.002
    step = net.train_step(lr=lr, weight_decay=0.001, ignored_label=25, use_graph=False)
    o64 = copy.deepcopy(ora).double().train()
    opt = torch.optim.Adam(o64.parameters(), lr=lr, weight_decay=0.001)
    rng = np.random.default_rng(3)
    for k in range(4):
        batch = workloads.mp3d_like_batch(int(rng.integers(3, 9)), seed=50 + k)
        y = batch["rooms"].y
        step(batch.to(DEV), y.to(DEV))
        loss_eng = step.loss()
        b64 = batch.to("cpu")
        for t in b64.node_types:
            b64[t].x = b64[t].x.double()
        opt.zero_grad()
        loss = o64.loss(o64(b64), y, y != 25)
        loss.backward()
        opt.step()
        assert abs(loss_eng - float(loss)) < 5e-5 * max(1.0, abs(float(loss))), (k, loss_eng, float(loss))
    ref = dict(o64.named_parameters())
    for name, p in net.named_parameters():
        if ref[name].grad is None:
            continue
        # Adam's first steps move every weight by ~lr whatever the gradient's size: compare in units of lr
        assert float((p.detach().cpu().double() - ref[name].detach()).abs().max()) < 0.05 * lr, name
    assert {k for k, on in net.native()._agg_first.items() if on} == {(0, 2), (1, 2), (2, 2), (0, 3), (1, 3)}


def test_aggregate_first_on_the_htree_network(monkeypatch):
    """10 edge types, 4 node types, LeafPool: the forced choice takes one conv per source type and layer; parity at 1e-5"""
    monkeypatch.setenv("HMP_AGG_FIRST", "1")
    dims = {"object": 306, "room": 6, "object-room": 6, "room-room": 6, "object_virtual": 306, "room_virtual": 6}
    kw = dict(input_dim_dict=dims, output_dim=26, conv_block="GraphSAGE", hidden_dim=32, num_layers=3, disable_initialization=True, dropout=0.0)
    ora, net = pair(kw, seed=7, cls=(omodels.HeterogeneousNeuralTreeNetwork, HeterogeneousNeuralTreeNetwork))
    batch = workloads.htree_batch(5, seed=61)
    _, _, n = check_against_oracle(ora, net, batch, label_type="room_virtual", train=False)
    assert n >= 20 and sum(net.native()._agg_first.values()) >= 6
