"""GPU parity of K3 (GAT edge softmax + aggregation) and of the GAT / GAT_edge networks against the oracle.

Tolerance: 1e-5 (atol + rtol) against the oracle evaluated in float64 (north_star).
"""
import copy
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import _lib, workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork, HomogeneousNetwork  # noqa: E402
from hydra_gnn_amd.data import Data  # noqa: E402
from oracle import models as omodels  # noqa: E402
from oracle import pyg_ref  # noqa: E402
from test_gpu_ops import build_plan, rand_edges  # noqa: E402

ATOL, RTOL = 1e-5, 1e-5
DEV = "cuda:0"


def gat_reference(h_src, a_src, a_dst, edge_attr, v_edge, ei, n_dst, self_loops):
    """float64 restatement of GATConv steps 3-7 for given projected inputs."""
    if self_loops:
        n = min(h_src.size(0), n_dst)
        keep = ei[0] != ei[1]
        ei = ei[:, keep]
        loop = torch.arange(n)
        ei = torch.cat([ei, torch.stack([loop, loop])], 1)
        if edge_attr is not None:
            edge_attr = torch.cat([edge_attr[keep], edge_attr.new_zeros(n, edge_attr.size(1))], 0)
    j, i = ei[0], ei[1]
    raw = a_src[j] + a_dst[i]
    if edge_attr is not None:
        raw = raw + edge_attr @ v_edge
    e = torch.nn.functional.leaky_relu(raw, 0.2)
    alpha = pyg_ref.segment_softmax(e, i, n_dst)
    return pyg_ref.scatter_sum(alpha.unsqueeze(-1) * h_src[j], i, n_dst)


@pytest.mark.parametrize("H,Cc,edge,loops,n_src,n_dst,E", [
    (1, 6, False, False, 40, 30, 200), (3, 64, True, True, 50, 50, 300), (4, 128, False, True, 64, 64, 500),
    (4, 26, True, False, 70, 20, 400), (8, 16, True, True, 33, 33, 100), (2, 256, False, False, 10, 12, 0),
])
def test_gat_unit_forward_backward(H, Cc, edge, loops, n_src, n_dst, E):
    lib = _lib.require_device()
    rng = np.random.default_rng(H * 100 + Cc)
    ei = rand_edges(rng, E, n_src, n_dst)
    if loops and E:
        ei[1, :5] = ei[0, :5]  # make sure genuine self loops exist and get removed
    p = build_plan(ei.to(DEV), n_src, n_dst)
    Cp = (Cc + 3) // 4 * 4
    f = lambda *s: torch.from_numpy(rng.normal(0, 1, size=s).astype(np.float32))
    h = f(n_src, H, Cc); a_s = f(n_src, H); a_d = f(n_dst, H)
    ea = f(max(E, 1), 3)[:E] if edge else None
    ve = f(3, H) if edge else None
    gout = f(n_dst, H * Cc)

    h_pad = torch.zeros(n_src, H, Cp); h_pad[:, :, :Cc] = h
    h_dev = h_pad.view(n_src, H * Cp).contiguous().to(DEV)
    as_dev = torch.zeros(n_src, 8); as_dev[:, :H] = a_s; as_dev = as_dev.to(DEV)
    ad_dev = torch.zeros(n_dst, 8); ad_dev[:, :H] = a_d; ad_dev = ad_dev.to(DEV)
    ea_dev = ea.contiguous().to(DEV) if edge and E else None
    ve_dev = None
    if edge:
        ve8 = torch.zeros(3, 8); ve8[:, :H] = ve; ve_dev = ve8.to(DEV)
    n_loop = min(n_src, n_dst) if loops else 0
    smax = torch.zeros(n_dst, 8, device=DEV); sden = torch.zeros(n_dst, 8, device=DEV)
    ldo = (H * Cc + 3) // 4 * 4
    out = torch.zeros(n_dst, ldo, device=DEV)
    args = _lib.GatArgs(H, Cc, int(loops), 3 if edge else 0, 0.0, 0, 0, 0)
    ptr = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib.hmp_gat_fwd(h_dev.data_ptr(), H * Cp, as_dev.data_ptr(), 8, ad_dev.data_ptr(), 8, ptr(ea_dev), ptr(ve_dev),
                               p["plan"], args, smax.data_ptr(), sden.data_ptr(), out.data_ptr(), ldo, _lib.stream_ptr()))

    h64, as64, ad64 = (t.double().requires_grad_(True) for t in (h, a_s, a_d))
    ve64 = ve.double().requires_grad_(True) if edge else None
    ref = gat_reference(h64, as64, ad64, ea.double() if edge else None, ve64, ei, n_dst, loops).reshape(n_dst, H * Cc)
    torch.testing.assert_close(out[:, : H * Cc].cpu().double(), ref.detach(), atol=ATOL, rtol=RTOL)

    ref.backward(gout.double())
    g_dev = torch.zeros(n_dst, ldo); g_dev[:, : H * Cc] = gout; g_dev = g_dev.to(DEV)
    alpha_drop = torch.zeros(E + n_loop + 1, 8, device=DEV); dlogit = torch.zeros(E + n_loop + 1, 8, device=DEV)
    dl_orig = torch.zeros(max(E, 1), 8, device=DEV) if edge else None
    g_h = torch.zeros(n_src, H * Cp, device=DEV); g_as = torch.zeros(n_src, 8, device=DEV); g_ad = torch.zeros(n_dst, 8, device=DEV)
    _lib.check(lib.hmp_gat_bwd(g_dev.data_ptr(), ldo, h_dev.data_ptr(), H * Cp, as_dev.data_ptr(), 8, ad_dev.data_ptr(), 8,
                               ptr(ea_dev), ptr(ve_dev), p["plan"], args, smax.data_ptr(), sden.data_ptr(), alpha_drop.data_ptr(),
                               dlogit.data_ptr(), ptr(dl_orig), g_h.data_ptr(), H * Cp, g_as.data_ptr(), 8, g_ad.data_ptr(), 8,
                               _lib.stream_ptr()))
    torch.testing.assert_close(g_h.view(n_src, H, Cp)[:, :, :Cc].cpu().double(), h64.grad, atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(g_as[:, :H].cpu().double(), as64.grad, atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(g_ad[:, :H].cpu().double(), ad64.grad, atol=ATOL, rtol=RTOL)
    if edge and E:
        g_ve = ea.double().t() @ dl_orig[:, :H].cpu().double()
        torch.testing.assert_close(g_ve, ve64.grad, atol=ATOL, rtol=RTOL)


def gat_pair(conv_block, hidden, heads, concats, dropout=0.0, seed=0, in_dims=None):
    torch.manual_seed(seed)
    if in_dims is None:
        in_dims = {"objects": 306, "rooms": 6} if conv_block == "GAT" else {"objects": 303, "rooms": 3}
    kw = dict(input_dim_dict=in_dims, output_dim=26, conv_block=conv_block, GAT_hidden_dims=hidden, GAT_heads=heads,
              GAT_concats=concats, dropout=dropout)
    ora = omodels.HeterogeneousNetwork(**kw)
    net = HeterogeneousNetwork(**kw)
    # the oracle's GATConv bias starts at zero; randomise it so the bias path is exercised
    with torch.no_grad():
        for n_, p_ in ora.named_parameters():
            if n_.endswith(".bias"):
                p_.uniform_(-0.1, 0.1)
    net.load_state_dict(ora.state_dict(), strict=True)
    return ora, net.to(DEV)


def oracle_run(ora, batch):
    o64 = copy.deepcopy(ora).double()
    o64.eval()
    b64 = batch.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    for et in b64.edge_types:
        if "edge_attr" in b64[et]:
            b64[et].edge_attr = b64[et].edge_attr.double()
    pred = o64(b64)
    y = batch["rooms"].y
    loss = o64.loss(pred, y, y != 25)
    loss.backward()
    return o64, pred.detach(), loss.detach()


def check_model(ora, net, batch):
    o64, pred_ref, loss_ref = oracle_run(ora, batch)
    net.eval()
    pred = net(batch.to(DEV))
    torch.testing.assert_close(pred.cpu().double(), pred_ref, atol=ATOL, rtol=RTOL)
    y = batch["rooms"].y.to(DEV)
    loss = net.loss(pred, y, y != 25)
    torch.testing.assert_close(loss.cpu().double(), loss_ref, atol=ATOL, rtol=RTOL)
    loss.backward()
    og = dict(o64.named_parameters())
    for name, p in net.named_parameters():
        ref = og[name].grad
        if ref is None:
            assert p.grad is None, f"{name}: unexpected gradient"
            continue
        assert p.grad is not None, f"{name}: missing gradient"
        torch.testing.assert_close(p.grad.cpu().double(), ref, atol=ATOL, rtol=RTOL, msg=lambda m: f"{name}: {m}")


@pytest.mark.parametrize("hidden,heads,concats", [
    ([16, 16], [2, 2, 2], [True, True, False]),
    ([64, 64], [3, 3, 3], [False, False, False]),       # the shipped MP3D shape (config/mp3d/baseline_gt60.yaml)
    ([128, 128], [4, 4, 4], [True, True, False]),       # BASELINE config 3
    ([8], [1, 1], [True, False]),
])
def test_hetero_gat_parity(hidden, heads, concats):
    ora, net = gat_pair("GAT", hidden, heads, concats)
    check_model(ora, net, workloads.mp3d_like_batch(5, seed=21))


@pytest.mark.parametrize("hidden,heads,concats", [
    ([64, 64], [3, 3, 3], [False, False, False]),
    ([32, 32], [4, 4, 4], [True, True, False]),
])
def test_hetero_gat_edge_parity(hidden, heads, concats):
    ora, net = gat_pair("GAT_edge", hidden, heads, concats)
    check_model(ora, net, workloads.mp3d_like_batch(5, seed=22, relative_pos=True))


def make_gat_replay(net, batch):
    """dropout_fn for the oracle that replays the keep-masks the ENGINE drew in its last training forward: feature dropout per
    (layer, node type) and attention dropout per (layer, conv).  The engine numbers an attention element (pos, h) in an
    [E + n_loop, 8] tensor, pos = position of the edge in the destination-sorted (stable) list, self loops at E + node
    (include/hydra_mp.h, hmp_gat_args); the oracle's list is [original edges without self loops..., loop 0, loop 1, ...]."""
    lib = _lib.require_device()
    nt = {"objects": 0, "rooms": 1}
    ets = list(omodels.EDGE_TYPES)

    def mask(stream, pp, n, F):
        m = torch.zeros(n * F, dtype=torch.uint8, device=DEV)
        _lib.check(lib.hmp_dropout_mask(net._seed, net._rng_step, stream, pp, n, F, m.data_ptr(), _lib.stream_ptr()))
        return m.view(n, F).cpu()

    def replay(x, pp, training, tag):
        if not training or pp == 0:
            return x
        if not tag.endswith(".alpha"):
            layer, t = tag[1:].split(".", 1)
            keep = mask(int(layer) * 8 + nt[t], pp, x.size(0), x.size(1))
            return x * keep.to(x.dtype) / (1.0 - pp)
        layer, name = tag[1:-len(".alpha")].split(".", 1)
        et = tuple(name.split("__"))
        ei = batch[et].edge_index
        E = ei.size(1)
        loops = et[0] == et[2]
        n_loop = min(batch[et[0]].x.size(0), batch[et[2]].x.size(0)) if loops else 0
        m = mask(1000 + int(layer) * 16 + ets.index(et), pp, E + n_loop, 8)
        order = torch.sort(ei[1], stable=True).indices
        pos_of = torch.empty(E, dtype=torch.int64)
        pos_of[order] = torch.arange(E)
        rows = torch.cat([pos_of[ei[0] != ei[1]], E + torch.arange(n_loop)]) if loops else pos_of
        assert rows.numel() == x.size(0)
        keep = m[rows][:, : x.size(1)]
        return x * keep.to(x.dtype) / (1.0 - pp)

    return replay


@pytest.mark.parametrize("block,hidden,heads,concats", [
    ("GAT", [128, 128], [4, 4, 4], [True, True, False]),        # BASELINE config 3: the timed path
    ("GAT_edge", [128, 128], [4, 4, 4], [True, True, False]),
    ("GAT_edge", [64, 64], [3, 3, 3], [False, False, False]),   # the shipped MP3D shape (config/mp3d/baseline_gt60.yaml)
])
def test_gat_training_mode_parity_with_replayed_masks(block, hidden, heads, concats):
    """Training mode as train_mp3d.py runs it (dropout 0.25 on the attention coefficients AND on the ELU outputs): the oracle
    replays the engine's Philox keep-masks; logits, loss and every gradient at 1e-5.  (models/utils.py:31-87)"""
    training_mode_parity(block, hidden, heads, concats, workloads.mp3d_like_batch(6, seed=29, relative_pos=(block == "GAT_edge")))


@pytest.mark.parametrize("block", ["GAT", "GAT_edge"])
def test_gat_training_mode_parity_at_the_bench_size(block):
    """the same check on the batch `bench.py --config 3 [--gat-edge]` times (B = 64, ~4 k objects, ~27 k edges, 4 heads x
    128): the 128x128-tile projections, the split weight gradients and the multi-round gat_* launches that 6 graphs never reach."""
    batch = workloads.config3_batch(64, edge=(block == "GAT_edge"))
    assert batch["objects"].x.size(0) > 3000
    training_mode_parity(block, [128, 128], [4, 4, 4], [True, True, False], batch)


def test_gat_with_side_stream_branches_orders_pack_before_the_projection(monkeypatch):
    """HMP_BRANCH=1 forks a side stream for pack + layer-0 projection; a GAT front launch (plan + pack on the main
    stream) must not run next to it.  Same results as the single-stream sequence, bit for bit."""
    batch = workloads.config3_batch(6)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("HMP_BRANCH", mode)
        _, net = gat_pair("GAT", [16, 16], [2, 2, 2], [True, True, False], seed=3)
        net.eval()
        gb = batch.to(DEV)
        outs = [net(gb).detach().clone() for _ in range(4)]
        assert all(torch.equal(o, outs[0]) for o in outs)
        res[mode] = outs[0]
    monkeypatch.delenv("HMP_BRANCH")
    assert torch.equal(res["0"], res["1"])


def training_mode_parity(block, hidden, heads, concats, batch):
    p = 0.25
    ora, net = gat_pair(block, hidden, heads, concats, dropout=p)
    net.train()
    pred = net(batch.to(DEV))
    replay = make_gat_replay(net, batch)
    ora.dropout_fn = replay
    for m in ora.modules():
        if isinstance(m, pyg_ref.GATConv):
            m.dropout_fn = replay
    o64 = copy.deepcopy(ora).double()
    o64.train()
    b64 = batch.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    for et in b64.edge_types:
        if "edge_attr" in b64[et]:
            b64[et].edge_attr = b64[et].edge_attr.double()
    y = batch["rooms"].y
    pred_ref = o64(b64)
    loss_ref = o64.loss(pred_ref, y, y != 25)
    loss_ref.backward()
    torch.testing.assert_close(pred.cpu().double(), pred_ref.detach(), atol=ATOL, rtol=RTOL)
    yg = y.to(DEV)
    loss = net.loss(pred, yg, yg != 25)
    torch.testing.assert_close(loss.detach().cpu().double(), loss_ref.detach(), atol=ATOL, rtol=RTOL)
    loss.backward()
    og = dict(o64.named_parameters())
    n = 0
    for name, q in net.named_parameters():
        ref = og[name].grad
        if ref is None:
            assert q.grad is None, f"{name}: unexpected gradient"
            continue
        assert q.grad is not None, f"{name}: missing gradient"
        torch.testing.assert_close(q.grad.cpu().double(), ref, atol=ATOL, rtol=RTOL, msg=lambda m: f"{name}: {m}")
        n += 1
    assert n > 20
    # the masks really acted: the eval-mode logits differ
    net.eval()
    assert not torch.allclose(net(batch.to(DEV)), pred, atol=1e-3)


def test_gat_training_mode_runs_and_is_deterministic():
    _, net = gat_pair("GAT_edge", [32, 32], [2, 2, 2], [True, True, False], dropout=0.4)
    batch = workloads.mp3d_like_batch(4, seed=23, relative_pos=True).to(DEV)
    y = batch["rooms"].y
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=5, use_graph=False)
    losses = []
    for _ in range(8):
        step(batch, y)
        losses.append(step.loss())
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]
    _, net2 = gat_pair("GAT_edge", [32, 32], [2, 2, 2], [True, True, False], dropout=0.4)
    step2 = net2.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=5, use_graph=True)
    losses2 = []
    for _ in range(8):
        step2(batch, y)
        losses2.append(step2.loss())
    assert losses == losses2  # same seed => same Philox masks, eager and graph replay agree bit for bit


def test_homogeneous_gat_parity():
    torch.manual_seed(0)
    kw = dict(input_dim=6, output_dim=15, conv_block="GAT", GAT_hidden_dims=[16], GAT_heads=[2, 2], GAT_concats=[True, False], dropout=0.0)
    ora = omodels.HomogeneousNetwork(**kw)
    net = HomogeneousNetwork(**kw)
    net.load_state_dict(ora.state_dict(), strict=True)
    net = net.to(DEV).eval()
    rng = np.random.default_rng(1)
    n = 9
    x = torch.from_numpy(rng.normal(size=(n, 6)).astype(np.float32))
    ei = torch.tensor([[1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 2, 3], [0, 0, 0, 0, 0, 0, 0, 0, 2, 1, 3, 2]])
    room_mask = torch.zeros(n, dtype=torch.bool); room_mask[0] = True
    d = Data(x=x, edge_index=ei, room_mask=room_mask, y=torch.zeros(n, dtype=torch.int64))
    o64 = copy.deepcopy(ora).double().eval()
    ref = o64(Data(x=x.double(), edge_index=ei, room_mask=room_mask))
    out = net(d.to(DEV))
    torch.testing.assert_close(out.cpu().double(), ref.detach(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("block", ["GAT", "GAT_edge"])
def test_gat_front_launch_of_plan_and_pack_is_bit_identical(monkeypatch, block):
    """GAT nets: plan parts and pack blocks share one front launch (no projection role -- its operand is the pack's output), the
    link pass follows; HMP_FRONT=0 keeps pack, plan and link as separate launches.  Same plan, same packed operands: predictions,
    gradients and three optimiser steps agree to the last bit."""
    batch = workloads.config3_batch(6, edge=(block == "GAT_edge"))
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("HMP_FRONT", mode)
        _, net = gat_pair(block, [16, 16], [2, 2, 2], [True, True, False], seed=3)
        net.eval()
        gb = batch.to(DEV)
        pred = net(gb)
        y = gb["rooms"].y
        net.loss(pred, y, y != 25).backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        assert net.native().read_state()[1] == 0
        net.train()
        step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, use_graph=False)
        for _ in range(3):
            step(gb, y)
        res[mode] = (pred.detach().clone(), grads, step.loss(), {k: p.detach().clone() for k, p in net.named_parameters()})
    monkeypatch.delenv("HMP_FRONT")
    assert torch.equal(res["1"][0], res["0"][0])
    for k, g in res["0"][1].items():
        assert torch.equal(res["1"][1][k], g), k
    assert res["1"][2] == res["0"][2]
    for k, p0 in res["0"][3].items():
        assert torch.equal(res["1"][3][k], p0), k
