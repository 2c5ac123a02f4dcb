"""GPU parity of the native network executor against the oracle models (same state_dict, same inputs).

Tolerance (north_star): activations / logits within 1e-5 (atol + rtol) of the oracle; the oracle is evaluated
in float64 from the same fp32 parameters and inputs, gradients are compared the same way.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import _lib, workloads  # noqa: E402
from hydra_gnn_amd.data import HeteroData, collate  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402
from oracle import models as omodels  # noqa: E402

ATOL, RTOL = 1e-5, 1e-5
DEV = "cuda:0"


def small_batch(n_graphs=6, seed=11):
    return workloads.mp3d_like_batch(n_graphs, seed)


def sage_pair(hidden=64, layers=3, dropout=0.0, seed=0):
    torch.manual_seed(seed)
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=hidden,
              num_layers=layers, dropout=dropout)
    ora = omodels.HeterogeneousNetwork(**kw)
    net = HeterogeneousNetwork(**kw)
    net.load_state_dict(ora.state_dict(), strict=True)
    return ora, net.to(DEV)


def oracle_run(ora, batch, train=False):
    o64 = copy.deepcopy(ora).double()
    o64.train(train)
    b64 = batch.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    pred = o64(b64)
    y = batch["rooms"].y
    loss = o64.loss(pred, y, y != 25)
    loss.backward()
    return o64, pred.detach(), loss.detach()


def assert_grads_close(net, o64, scale=1.0):
    og = dict(o64.named_parameters())
    for name, p in net.named_parameters():
        ref = og[name].grad
        if ref is None:
            assert p.grad is None, f"{name}: engine produced a gradient the reference would not"
            continue
        assert p.grad is not None, f"{name}: missing gradient"
        torch.testing.assert_close(p.grad.cpu().double() * scale, ref, atol=ATOL, rtol=RTOL, msg=lambda m: f"{name}: {m}")


@pytest.mark.parametrize("hidden,layers", [(64, 3), (32, 2), (128, 4)])
def test_hetero_sage_forward_backward_parity(hidden, layers):
    ora, net = sage_pair(hidden, layers)
    batch = small_batch()
    o64, pred_ref, loss_ref = oracle_run(ora, batch)
    net.eval()
    pred = net(batch.to(DEV))
    assert pred.shape == pred_ref.shape
    torch.testing.assert_close(pred.cpu().double(), pred_ref, atol=ATOL, rtol=RTOL)
    y = batch["rooms"].y.to(DEV)
    loss = net.loss(pred, y, y != 25)
    torch.testing.assert_close(loss.cpu().double(), loss_ref, atol=ATOL, rtol=RTOL)
    loss.backward()
    assert_grads_close(net, o64)


def test_config2_full_batch_parity():
    ora, net = sage_pair(64, 3)
    batch = workloads.config2_batch(32)
    o64, pred_ref, loss_ref = oracle_run(ora, batch)
    net.eval()
    pred = net(batch.to(DEV))
    torch.testing.assert_close(pred.cpu().double(), pred_ref, atol=ATOL, rtol=RTOL)
    y = batch["rooms"].y.to(DEV)
    loss = net.loss(pred, y, y != 25)
    loss.backward()
    assert_grads_close(net, o64)


def test_empty_and_ragged_edge_types():
    """rooms_to_rooms empty, one graph with a single room / single object, isolated objects."""
    g = HeteroData()
    rng = np.random.default_rng(3)
    g["objects"].x = torch.from_numpy(rng.normal(size=(5, 306)).astype(np.float32))
    g["rooms"].x = torch.from_numpy(rng.normal(size=(2, 6)).astype(np.float32))
    g["rooms"].y = torch.tensor([3, 25])
    g["objects", "objects_to_objects", "objects"].edge_index = torch.tensor([[0, 1, 1], [1, 0, 0]])  # duplicate edge
    g["rooms", "rooms_to_rooms", "rooms"].edge_index = torch.empty((2, 0), dtype=torch.int64)
    g["rooms", "rooms_to_objects", "objects"].edge_index = torch.tensor([[0, 0, 1], [0, 1, 4]])  # objects 2,3 orphaned
    g["objects", "objects_to_rooms", "rooms"].edge_index = torch.tensor([[0, 1, 4], [0, 0, 1]])
    batch = collate([g, g])
    ora, net = sage_pair(16, 3)
    o64, pred_ref, _ = oracle_run(ora, batch)
    net.eval()
    pred = net(batch.to(DEV))
    torch.testing.assert_close(pred.cpu().double(), pred_ref, atol=ATOL, rtol=RTOL)
    y = batch["rooms"].y.to(DEV)
    net.loss(pred, y, y != 25).backward()
    assert_grads_close(net, o64)


def test_dropout_training_parity_with_replayed_masks():
    """Training-mode parity: the oracle replays the engine's Philox keep-masks (hmp_dropout_mask)."""
    import ctypes as C

    lib = _lib.require_device()
    p = 0.25
    ora, net = sage_pair(64, 3, dropout=p)
    batch = small_batch()
    net.train()
    pred = net(batch.to(DEV))  # rng_step becomes 1
    nt = {"objects": 0, "rooms": 1}

    def replay(x, pp, training, tag):
        if not training or pp == 0:
            return x
        layer, t = tag[1:].split(".", 1)
        n, F = x.shape
        m = torch.zeros(n * F, dtype=torch.uint8, device=DEV)
        _lib.check(lib.hmp_dropout_mask(net._seed, net._rng_step, int(layer) * 8 + nt[t], pp, n, F, m.data_ptr(), _lib.stream_ptr()))
        keep = m.view(n, F).cpu().to(x.dtype)
        return x * keep / (1.0 - pp)

    ora.dropout_fn = replay
    o64, pred_ref, _ = oracle_run(ora, batch, train=True)
    torch.testing.assert_close(pred.cpu().double(), pred_ref, atol=ATOL, rtol=RTOL)
    y = batch["rooms"].y.to(DEV)
    net.loss(pred, y, y != 25).backward()
    assert_grads_close(net, o64)


def test_state_dict_round_trip_and_pyg24_keys():
    ora, net = sage_pair(32, 3)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    # PyG >= 2.4 ModuleDict spelling
    sd24 = {}
    for k, v in sd.items():
        parts = k.split(".")
        if len(parts) > 3 and parts[2] == "convs":
            parts[3] = "<" + parts[3].replace("__", "___") + ">"
        sd24[".".join(parts)] = v
    _, net2 = sage_pair(32, 3, seed=5)
    net2.load_state_dict(sd24, strict=True)
    batch = small_batch(3)
    net.eval(); net2.eval()
    assert torch.equal(net(batch.to(DEV)), net2(batch.to(DEV)))


def test_fused_train_step_matches_oracle_adam():
    """5 fused native steps (plan + fwd + CE + bwd + Adam, hipGraph replay) == oracle + torch.optim.Adam."""
    ora, net = sage_pair(64, 3, dropout=0.0)
    batch = workloads.config2_batch(8)
    o64 = copy.deepcopy(ora).double()
    opt = torch.optim.Adam(o64.parameters(), lr=0.002, weight_decay=0.001)
    b64 = batch.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    y = batch["rooms"].y
    losses_ref = []
    # Adam's update m/(sqrt(v)+eps) is ill-conditioned where |g| ~ eps: a 1e-9 difference in a 1e-8 gradient moves
    # the parameter by ~lr.  Those elements are excluded from the element-wise comparison (and counted).
    tiny = {n: torch.zeros_like(p, dtype=torch.bool) for n, p in o64.named_parameters()}
    for _ in range(5):
        opt.zero_grad()
        loss = o64.loss(o64(b64), y, y != 25)
        loss.backward()
        for n, p in o64.named_parameters():
            if p.grad is not None:
                tiny[n] |= p.grad.abs() < 1e-5
        opt.step()
        losses_ref.append(float(loss.detach()))
    for use_graph in (False, True):
        _, net = sage_pair(64, 3, dropout=0.0)
        step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, use_graph=use_graph)
        gb = batch.to(DEV)
        yg = y.to(DEV)
        losses = []
        for _ in range(5):
            step(gb, yg)
            losses.append(step.loss())
        np.testing.assert_allclose(losses, losses_ref, rtol=2e-5, atol=2e-5)
        ref = dict(o64.named_parameters())
        n_tiny = n_all = 0
        for name, p in net.named_parameters():
            ok = ~tiny[name]
            n_tiny += int(tiny[name].sum()); n_all += tiny[name].numel()
            # gradients agree to ~1e-6 absolute, Adam divides by |g|: allow lr * 1e-6/1e-5 per step on the rest,
            # and require the bulk of the tensor to agree tightly
            diff = (p.detach().cpu().double() - ref[name].detach()).abs()
            if bool(ok.any()):
                assert float(diff[ok].max()) <= 1e-3, f"{name} (graph={use_graph})"  # 10 % of the 5*lr travelled
                assert float((diff[ok] > 5e-5).double().mean()) < 0.01, f"{name} (graph={use_graph})"
            # ill-conditioned elements still cannot move further than 5 steps of size lr (+ weight decay drift)
            assert float((p.detach().cpu().double() - ref[name].detach()).abs().max()) <= 5 * 0.002 * 2.1
        assert n_tiny < 0.2 * n_all
        st, status = net.native().read_state()
        assert st == 5 and status == 0


def test_engine_is_deterministic():
    _, net = sage_pair(64, 3)
    batch = workloads.config2_batch(8).to(DEV)
    net.eval()
    a = net(batch)
    y = batch["rooms"].y
    net.loss(a, y, y != 25).backward()
    ga = [p.grad.clone() for p in net.parameters() if p.grad is not None]
    for p in net.parameters():
        p.grad = None
    b = net(batch)
    net.loss(b, y, y != 25).backward()
    gb = [p.grad for p in net.parameters() if p.grad is not None]
    assert torch.equal(a, b)
    assert all(torch.equal(u, v) for u, v in zip(ga, gb))


def test_cpu_tensors_are_rejected():
    _, net = sage_pair(16, 2)
    with pytest.raises(_lib.HydraMPError):
        net(small_batch(2))  # batch left on the CPU


# ---- two-headed task (classification_task 'all': output_dim_dict) -------------------------------------------------------------
def hetero_replay(net):
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


def two_head_check(ora, net, batch, b64):
    """forward (both outputs) and the gradients of a loss that uses both, against the oracle in float64"""
    net.train()
    pr, po = net(batch.to(DEV))
    ora.dropout_fn = hetero_replay(net)
    o64 = copy.deepcopy(ora).double().train()
    rr, ro = o64(b64)
    assert pr.shape == rr.shape and po.shape == ro.shape and po.shape[0] > 0 and pr.shape[1] != po.shape[1]
    torch.testing.assert_close(pr.detach().cpu().double(), rr.detach(), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(po.detach().cpu().double(), ro.detach(), atol=ATOL, rtol=RTOL)
    ((rr * rr).sum() / rr.shape[0] + ro.sum() / ro.shape[0]).backward()
    ((pr * pr).sum() / pr.shape[0] + po.sum() / po.shape[0]).backward()
    og = dict(o64.named_parameters())
    for name, p in net.named_parameters():
        assert (p.grad is None) == (og[name].grad is None), name
        if p.grad is not None:
            torch.testing.assert_close(p.grad.cpu().double(), og[name].grad, atol=ATOL, rtol=RTOL, msg=lambda m: f"{name}: {m}")
    return net, o64


@pytest.mark.parametrize("block", ["GraphSAGE", "GAT"])
def test_hetero_two_head_task_parity(block):
    """``HeterogeneousNetwork(output_dim_dict=...)`` returns (rooms, objects) after activation + dropout on the final states
    (reference heterogeneous_network.py:123-135): the executor's second readout (hmp_net_aux_output / hmp_net_backward2); every
    last-layer conv is live, the two outputs have different widths"""
    torch.manual_seed(2)
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim_dict={"rooms": 26, "objects": 35}, conv_block=block,
              hidden_dim=32, num_layers=3, dropout=0.25)
    if block == "GAT":  # attention-dropout masks are not replayed here (test_gpu_gat.py covers them)
        kw.update(GAT_hidden_dims=[16, 16], GAT_heads=[2, 2, 2], GAT_concats=[True, True, False], dropout=0.0)
    ora = omodels.HeterogeneousNetwork(**kw)
    net = HeterogeneousNetwork(**kw)
    net.load_state_dict(ora.state_dict(), strict=True)
    batch = small_batch(5, seed=17)
    b64 = batch.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    net, o64 = two_head_check(ora, net.to(DEV), batch, b64)
    last = f"convs.{net.num_layers - 1}."
    for name, p in net.named_parameters():  # no last-layer conv is dead when both final states are outputs
        assert p.grad is not None or not name.startswith(last), name
    with pytest.raises(NotImplementedError):
        net.train_step(lr=1e-3)


def test_hetero_two_head_one_output_unused():
    """a loss over ONE of the two outputs: the other's gradient enters as zeros (autograd materialises it), in either order"""
    torch.manual_seed(3)
    kw = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim_dict={"rooms": 7, "objects": 9}, conv_block="GraphSAGE",
              hidden_dim=16, num_layers=2, dropout=0.0)
    ora = omodels.HeterogeneousNetwork(**kw)
    net = HeterogeneousNetwork(**kw)
    net.load_state_dict(ora.state_dict(), strict=True)
    net = net.to(DEV).eval()
    batch = small_batch(3, seed=19)
    b64 = batch.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    for which in (0, 1):
        o64 = copy.deepcopy(ora).double().eval()
        o64(b64)[which].square().mean().backward()
        net.zero_grad(set_to_none=True)
        net(batch.to(DEV))[which].square().mean().backward()
        og = dict(o64.named_parameters())
        for name, p in net.named_parameters():
            ref = og[name].grad
            got = p.grad.cpu().double() if p.grad is not None else None
            if ref is None:  # the executor's liveness is static: a conv that only feeds the unused output gets exact zeros
                assert got is None or float(got.abs().max()) == 0.0, name
            else:
                torch.testing.assert_close(got, ref, atol=ATOL, rtol=RTOL, msg=lambda m: f"{name}: {m}")
