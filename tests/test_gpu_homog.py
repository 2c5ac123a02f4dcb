"""GPU parity of the homogeneous GCN / GIN (+ BatchNorm) family: the operators of csrc/homog.hip
against dense float64 restatements, and the models (``baseline_GCN`` / ``baseline_GIN`` / ``htree_GCN`` / ``htree_GIN`` shapes)
against the oracle with the same state_dict.

Tolerance (north_star): fp32 results within 1e-5 (atol + rtol) of the float64 oracle (gradient sums over all rows: 1e-4);
deg^-1/2 of the integer degrees within 2 ulp (device rsqrt).
Parity status of the oracle itself: unpinned (oracle/pyg_ref.py header) -- its GCN / GIN restatements carry hand-derived
known answers in tests/test_oracle_kat.py.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import _lib, ops, workloads  # noqa: E402
from hydra_gnn_amd.data import collate_homogeneous  # noqa: E402
from hydra_gnn_amd.models import HomogeneousNetwork, HomogeneousNeuralTreeNetwork  # noqa: E402
from oracle import models as omodels  # noqa: E402

ATOL, RTOL = 1e-5, 1e-5
DEV = "cuda:0"


def rand_graph(n, E, seed, self_loops=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, max(n, 1), (2, E), generator=g)
    if not self_loops:
        ei = ei[:, ei[0] != ei[1]]
    return ei


def dense_adj(ei, n):
    """A[i, j] = number of edges j -> i (float64)"""
    A = torch.zeros(n, n, dtype=torch.float64)
    A.index_put_((ei[1], ei[0]), torch.ones(ei.size(1), dtype=torch.float64), accumulate=True)
    return A


# ---- operators ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,E,F", [(1, 0, 3), (5, 9, 6), (27, 286, 15), (700, 5000, 64), (300, 4000, 130), (64, 900, 300)])
def test_gcn_norm_and_propagate_both_directions(n, E, F):
    ei = rand_graph(n, E, seed=n + E)
    plan = ops.GraphPlan(ei.to(DEV), n)
    A = dense_adj(ei, n)
    A.fill_diagonal_(0.0)  # existing loops are replaced ...
    deg = A.sum(1) + 1.0  # ... by exactly one
    dinv = plan.dinv.cpu()[:n]
    torch.testing.assert_close(dinv.double(), deg.pow(-0.5), atol=0.0, rtol=2.5e-7)  # integer degrees; 2 ulp for the device rsqrt
    x = torch.randn(n, F, generator=torch.Generator().manual_seed(1))
    S = torch.diag(deg.pow(-0.5)) @ (A + torch.eye(n, dtype=torch.float64)) @ torch.diag(deg.pow(-0.5))
    xg = x.to(DEV).requires_grad_(True)
    out = ops.gcn_propagate(xg, plan)
    torch.testing.assert_close(out.detach().cpu().double(), S @ x.double(), atol=ATOL, rtol=RTOL)
    g = torch.randn(n, F, generator=torch.Generator().manual_seed(2))
    out.backward(g.to(DEV))
    torch.testing.assert_close(xg.grad.cpu().double(), S.t() @ g.double(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("n,E,F,eps", [(1, 0, 3, 0.0), (5, 9, 6, 0.25), (27, 286, 15, -0.5), (700, 5000, 64, 0.1), (64, 900, 300, 1.5)])
def test_gin_propagate_and_eps_gradient(n, E, F, eps):
    ei = rand_graph(n, E, seed=3 * n + E)
    plan = ops.GraphPlan(ei.to(DEV), n)
    A = dense_adj(ei, n)  # GIN keeps loops and multi-edges as they are
    x = torch.randn(n, F, generator=torch.Generator().manual_seed(1))
    xg = x.to(DEV).requires_grad_(True)
    e = torch.tensor([eps], device=DEV, requires_grad=True)
    out = ops.gin_propagate(xg, plan, e)
    M = A + (1.0 + eps) * torch.eye(n, dtype=torch.float64)
    torch.testing.assert_close(out.detach().cpu().double(), M @ x.double(), atol=ATOL, rtol=RTOL)
    g = torch.randn(n, F, generator=torch.Generator().manual_seed(2))
    out.backward(g.to(DEV))
    torch.testing.assert_close(xg.grad.cpu().double(), M.t() @ g.double(), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(e.grad.cpu().double(), (g.double() * x.double()).sum().reshape(1), atol=1e-4, rtol=RTOL)


def test_propagate_skips_edges_the_plan_dropped():
    """an out-of-range endpoint sets the plan's status bit and the edge takes no part (hydra_mp.h section 1)"""
    ei = torch.tensor([[0, 1, 7, 2], [1, 0, 1, 9]])
    plan = ops.GraphPlan(ei.to(DEV), 3)
    x = torch.tensor([[1.0], [2.0], [4.0]], device=DEV)
    out = ops.gin_propagate(x, plan, torch.zeros(1, device=DEV))
    torch.cuda.synchronize()
    assert int(plan.status.item()) & 1
    assert out.cpu().flatten().tolist() == [3.0, 3.0, 4.0]


@pytest.mark.parametrize("n,F,p", [(33, 15, 0.0), (500, 64, 0.25), (129, 130, 0.5)])
def test_bias_elu_drop_matches_replayed_mask(n, F, p):
    lib = _lib.require_device()
    g0 = torch.Generator().manual_seed(6)
    x, b, g = torch.randn(n, F, generator=g0), torch.randn(F, generator=g0), torch.randn(n, F, generator=g0)
    x[0, :4] = -b[:4]  # exact zeros before the activation: a kept zero must not read as "dropped"
    xg, bg = x.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.bias_act_drop(xg, bg, elu=True, p=p, seed=7, rng_step=2, rng_stream=24)
    y.backward(g.to(DEV))
    m = torch.ones(n * F, dtype=torch.uint8, device=DEV)
    if p > 0:
        _lib.check(lib.hmp_dropout_mask(7, 2, 24, p, n, F, m.data_ptr(), _lib.stream_ptr()))
    keep = m.view(n, F).cpu().double()
    x64, b64 = x.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.elu(x64 + b64) * keep / (1.0 - p)
    ref.backward(g.double())
    torch.testing.assert_close(y.detach().cpu().double(), ref.detach(), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(xg.grad.cpu().double(), x64.grad, atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(bg.grad.cpu().double(), b64.grad, atol=1e-4, rtol=RTOL)


@pytest.mark.parametrize("n,F,relu,p", [(7, 5, False, 0.0), (33, 15, True, 0.0), (500, 64, True, 0.25), (129, 130, True, 0.5)])
def test_bias_act_drop_matches_replayed_mask(n, F, relu, p):
    lib = _lib.require_device()
    g0 = torch.Generator().manual_seed(5)
    x, b, g = torch.randn(n, F, generator=g0), torch.randn(F, generator=g0), torch.randn(n, F, generator=g0)
    xg, bg = x.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.bias_act_drop(xg, bg, relu=relu, p=p, seed=99, rng_step=4, rng_stream=16)
    y.backward(g.to(DEV))
    m = torch.ones(n * F, dtype=torch.uint8, device=DEV)
    if p > 0:
        _lib.check(lib.hmp_dropout_mask(99, 4, 16, p, n, F, m.data_ptr(), _lib.stream_ptr()))
        assert abs(float(m.float().mean()) - (1 - p)) < 0.05
    keep = m.view(n, F).cpu().double()
    x64 = x.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    pre = x64 + b64
    ref = (torch.relu(pre) if relu else pre) * keep / (1.0 - p)
    ref.backward(g.double())
    torch.testing.assert_close(y.detach().cpu().double(), ref.detach(), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(xg.grad.cpu().double(), x64.grad, atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(bg.grad.cpu().double(), b64.grad, atol=1e-4, rtol=RTOL)


def test_dropout_without_relu_is_refused():
    with pytest.raises(_lib.HydraMPError):
        ops.bias_act_drop(torch.zeros(4, 4, device=DEV), None, relu=False, p=0.5)


@pytest.mark.parametrize("n,F", [(2, 3), (40, 64), (1000, 128), (333, 70)])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_matches_torch_batchnorm1d(n, F, training):
    g0 = torch.Generator().manual_seed(n + F)
    x = torch.randn(n, F, generator=g0) * 3 + 1.5
    g = torch.randn(n, F, generator=g0)
    ref = torch.nn.BatchNorm1d(F).double()
    with torch.no_grad():
        ref.weight.copy_(torch.rand(F, generator=g0) + 0.5), ref.bias.copy_(torch.randn(F, generator=g0))
        ref.running_mean.copy_(torch.randn(F, generator=g0)), ref.running_var.copy_(torch.rand(F, generator=g0) + 0.5)
    ref.train(training)
    gamma = ref.weight.detach().float().to(DEV).requires_grad_(True)
    beta = ref.bias.detach().float().to(DEV).requires_grad_(True)
    rm, rv = ref.running_mean.float().to(DEV), ref.running_var.float().to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.batch_norm(xg, gamma, beta, rm, rv, 0.1, 1e-5, training)
    y.backward(g.to(DEV))
    x64 = x.double().requires_grad_(True)
    yr = ref(x64)
    yr.backward(g.double())
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(xg.grad.cpu().double(), x64.grad, atol=ATOL, rtol=1e-4)
    torch.testing.assert_close(gamma.grad.cpu().double(), ref.weight.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(beta.grad.cpu().double(), ref.bias.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(rm.cpu().double(), ref.running_mean, atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(rv.cpu().double(), ref.running_var, atol=ATOL, rtol=RTOL)


def test_batchnorm_training_needs_two_rows():
    z = torch.zeros(1, 4, device=DEV)
    o = torch.ones(4, device=DEV)
    with pytest.raises(_lib.HydraMPError):
        ops.batch_norm(z, o, o.clone(), o.clone(), o.clone(), 0.1, 1e-5, True)


def test_project_matches_linear():
    g0 = torch.Generator().manual_seed(8)
    x, w, g = torch.randn(37, 6, generator=g0), torch.randn(15, 6, generator=g0), torch.randn(37, 15, generator=g0)
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ops.project(xg, wg)
    y.backward(g.to(DEV))
    torch.testing.assert_close(y.detach().cpu().double(), x.double() @ w.double().t(), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(xg.grad.cpu().double(), g.double() @ w.double(), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(wg.grad.cpu().double(), g.double().t() @ x.double(), atol=ATOL, rtol=RTOL)


def test_ops_refuse_cpu_tensors():
    with pytest.raises(_lib.HydraMPError):
        ops.project(torch.zeros(2, 2), torch.zeros(2, 2))
    with pytest.raises(_lib.HydraMPError):
        ops.GraphPlan(torch.zeros(2, 0, dtype=torch.int64), 2)


# ---- models ---------------------------------------------------------------------------------------------------------------
def stanford_batch(n_graphs, seed=1):
    rng = np.random.Generator(np.random.PCG64(workloads.BASE_SEED + seed))
    return collate_homogeneous([workloads.stanford_like_graph(rng, n_nodes=(2 if i == 0 else None)) for i in range(n_graphs)])


def pair(cls, ocls, **kw):
    torch.manual_seed(4)
    ora = ocls(**kw)
    with torch.no_grad():  # zero-initialised biases / unit BatchNorm would hide mistakes
        for name, p in ora.named_parameters():
            if name.endswith("bias") or "batch_norms" in name:
                p.add_(torch.randn_like(p) * 0.2)
            if name.endswith("eps"):
                p.fill_(0.3)
        for name, b in ora.named_buffers():
            if name.endswith("running_mean"):
                b.add_(torch.randn_like(b) * 0.2)
            if name.endswith("running_var"):
                b.mul_(torch.rand_like(b) + 0.5)
    net = cls(**kw)
    net.load_state_dict(ora.state_dict(), strict=True)
    assert list(net.state_dict()) == list(ora.state_dict())
    return ora, net.to(DEV)


def replay_fn(net):
    lib = _lib.require_device()

    def replay(x, pp, training, tag):
        if not training or pp == 0:
            return x
        layer = int(tag[1:].split(".", 1)[0])
        n, F = x.shape
        m = torch.zeros(n * F, dtype=torch.uint8, device=DEV)
        _lib.check(lib.hmp_dropout_mask(net._seed, net._rng_step, net._drop_stream(layer), pp, n, F, m.data_ptr(), _lib.stream_ptr()))
        return x * m.view(n, F).cpu().to(x.dtype) / (1.0 - pp)

    return replay


def check_model(ora, net, batch, train, label_key="y"):
    net.train(train)
    pred = net(batch.to(DEV))
    ora.dropout_fn = replay_fn(net)
    o64 = copy.deepcopy(ora).double()
    o64.train(train)
    b64 = batch.to("cpu")
    b64.x = b64.x.double()
    pred_ref = o64(b64)
    torch.testing.assert_close(pred.detach().cpu().double(), pred_ref.detach(), atol=ATOL, rtol=RTOL)
    y = batch.y[batch.room_mask]
    loss_ref = o64.loss(pred_ref, y)
    loss_ref.backward()
    loss = net.loss(pred, y.to(DEV))
    torch.testing.assert_close(loss.detach().cpu().double(), loss_ref.detach(), atol=ATOL, rtol=RTOL)
    loss.backward()
    og = dict(o64.named_parameters())
    for name, p in net.named_parameters():
        ref = og[name].grad
        if ref is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        torch.testing.assert_close(p.grad.cpu().double(), ref, atol=ATOL, rtol=1e-4, msg=lambda m: f"{name}: {m}")
    ob = dict(o64.named_buffers())
    for name, b in net.named_buffers():
        torch.testing.assert_close(b.cpu().double(), ob[name].double(), atol=ATOL, rtol=RTOL, msg=lambda m: f"{name}: {m}")
    return pred


@pytest.mark.parametrize("block", ["GCN", "GIN"])
@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("n_graphs", [2, 24])
def test_baseline_gcn_gin_parity(block, train, n_graphs):
    """config/Stanford3D/baseline_GCN.yaml / baseline_GIN.yaml: hidden 64, 3 layers, 6-d features, 15 room classes"""
    kw = dict(input_dim=6, output_dim=15, conv_block=block, hidden_dim=64, num_layers=3, dropout=0.25 if train else 0.0)
    ora, net = pair(HomogeneousNetwork, omodels.HomogeneousNetwork, **kw)
    pred = check_model(ora, net, stanford_batch(n_graphs), train)
    assert pred.shape == (n_graphs, 15)
    if block == "GIN":
        assert [int(b.module.num_batches_tracked) for b in net.batch_norms] == ([1, 1, 0] if train else [0, 0, 0])


GAT_KW = dict(GAT_hidden_dims=[16, 16], GAT_heads=[2, 2], GAT_concats=[True, False])


def two_head_check(ora, net, batch):
    net.train()
    pr, po = net(batch.to(DEV))
    ora.dropout_fn = replay_fn(net)
    o64 = copy.deepcopy(ora).double().train()
    b64 = batch.to("cpu")
    b64.x = b64.x.double()
    rr, ro = o64(b64)
    assert pr.shape == rr.shape and po.shape == ro.shape and po.shape[0] > 0
    torch.testing.assert_close(pr.detach().cpu().double(), rr.detach(), atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(po.detach().cpu().double(), ro.detach(), atol=ATOL, rtol=RTOL)
    (rr.sum() + (ro * ro).sum()).backward()
    (pr.sum() + (po * po).sum()).backward()
    og = dict(o64.named_parameters())
    for name, p in net.named_parameters():
        if og[name].grad is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        torch.testing.assert_close(p.grad.cpu().double(), og[name].grad, atol=1e-4, rtol=1e-4, msg=lambda m: f"{name}: {m}")


@pytest.mark.parametrize("block", ["GCN", "GIN", "GraphSAGE", "GAT"])
def test_two_head_task_parity(block):
    """classification_task 'all' (output_dim_dict, the semi-supervised Stanford job): activation + dropout after the last conv,
    then one Linear per node set; SAGE / GAT run the convs as the native program and the tail on the native operators"""
    kw = dict(input_dim=6, output_dim_dict={"room": 15, "object": 35}, conv_block=block, hidden_dim=32, num_layers=2, dropout=0.25)
    if block == "GAT":  # GATConv draws attention-dropout masks this file does not replay: elu + dropout is covered at op level
        kw.update(GAT_KW, dropout=0.0)
    ora, net = pair(HomogeneousNetwork, omodels.HomogeneousNetwork, **kw)
    two_head_check(ora, net, stanford_batch(9, seed=2))
    with pytest.raises(NotImplementedError):
        net.train_step(lr=1e-3)


@pytest.mark.parametrize("block", ["GIN", "GraphSAGE", "GAT"])
def test_two_head_task_parity_htree(block):
    """same on the homogeneous H-tree: pool first, then activation + dropout + heads over room_mask / object_mask
    (homogeneous_neural_tree_network.py:96-109)"""
    from test_gpu_htree import homogeneous_htree_batch

    kw = dict(input_dim=6, output_dim_dict={"room": 26, "object": 28}, conv_block=block, hidden_dim=32, num_layers=3,
              disable_initialization=False, dropout=0.25)
    if block == "GAT":
        kw.update(GAT_hidden_dims=[16, 16, 16], GAT_heads=[2, 2, 2], GAT_concats=[True, True, False], dropout=0.0)
    ora, net = pair(HomogeneousNeuralTreeNetwork, omodels.HomogeneousNeuralTreeNetwork, **kw)
    batch = homogeneous_htree_batch(4, seed=47)
    batch.x = batch.x[:, :6].contiguous()
    assert int(batch.object_mask.sum()) > 0
    two_head_check(ora, net, batch)


@pytest.mark.parametrize("block", ["GCN", "GIN"])
@pytest.mark.parametrize("init", [False, True])
def test_htree_gcn_gin_parity(block, init):
    """config/Stanford3D/htree_GCN.yaml / htree_GIN.yaml shapes: pre_mp GAT (native program) -> convs (ops) -> LeafPool (op);
    the H-tree loop applies no BatchNorm (homogeneous_neural_tree_network.py:86-94)."""
    from test_gpu_htree import homogeneous_htree_batch

    kw = dict(input_dim=6, output_dim=26, conv_block=block, hidden_dim=32, num_layers=4, disable_initialization=not init, dropout=0.25)
    ora, net = pair(HomogeneousNeuralTreeNetwork, omodels.HomogeneousNeuralTreeNetwork, **kw)
    batch = homogeneous_htree_batch(4, seed=43)
    batch.x = batch.x[:, :6].contiguous()
    net.train()
    pred = net(batch.to(DEV))
    ora.dropout_fn = replay_fn(net)
    o64 = copy.deepcopy(ora).double().train()
    b64 = batch.to("cpu")
    b64.x = b64.x.double()
    pred_ref = o64(b64)
    assert pred.shape == (int(batch.room_mask.sum()), 26)
    torch.testing.assert_close(pred.detach().cpu().double(), pred_ref.detach(), atol=ATOL, rtol=RTOL)
    y = batch.y[batch.room_mask]
    o64.loss(pred_ref, y, y != 25).backward()
    yg = y.to(DEV)
    net.loss(pred, yg, yg != 25).backward()
    og = dict(o64.named_parameters())
    for name, p in net.named_parameters():
        if og[name].grad is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        torch.testing.assert_close(p.grad.cpu().double(), og[name].grad, atol=ATOL, rtol=1e-4, msg=lambda m: f"{name}: {m}")
    if block == "GIN":
        assert [int(b.module.num_batches_tracked) for b in net.batch_norms] == [0, 0, 0, 0]


def test_gcn_trains_with_the_reference_loop():
    """base_training_job.py:202-216 as the reference writes it (zero_grad, forward, loss, backward, Adam.step)"""
    torch.manual_seed(0)
    net = HomogeneousNetwork(input_dim=6, output_dim=15, conv_block="GIN", hidden_dim=32, num_layers=3, dropout=0.1).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, weight_decay=1e-4)
    batch = stanford_batch(32, seed=5).to(DEV)
    y = batch.y[batch.room_mask]
    net.train()
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = net.loss(net(batch), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < 0.6 * losses[0]
    with pytest.raises(_lib.HydraMPError):
        net.train_step(lr=1e-3)
