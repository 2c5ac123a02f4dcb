"""Pins the oracle (CPU restatement of the PyG operators) with hand-derived known answers, float64
self-consistency and autograd gradcheck.  The reference holds no model-output vectors ("parity unpinned",
see oracle/pyg_ref.py), so these KATs are derived by hand from the published PyG 2.3.1 formulas
(SURVEY.md Appendix A)."""
import math

import pytest
import torch

from oracle import models as omodels
from oracle import pyg_ref as R

torch.set_default_dtype(torch.float32)


def test_sage_mean_known_answer():
    # 3 source nodes -> 2 destination nodes; edges (j->i): 0->0, 1->0, 2->0, 2->0 (duplicate counts twice), none ->1
    conv = R.SAGEConv((2, 1), 2)
    with torch.no_grad():
        conv.lin_l.weight.copy_(torch.tensor([[1.0, 0.0], [0.0, 2.0]]))
        conv.lin_l.bias.copy_(torch.tensor([0.5, -0.5]))
        conv.lin_r.weight.copy_(torch.tensor([[10.0], [100.0]]))
    x_src = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    x_dst = torch.tensor([[1.0], [2.0]])
    ei = torch.tensor([[0, 1, 2, 2], [0, 0, 0, 0]])
    out = conv((x_src, x_dst), ei)
    # mean over N(0) = (x0 + x1 + 2*x2)/4 = (14/4, 18/4) = (3.5, 4.5);  W_l m + b = (3.5+0.5, 9-0.5) = (4, 8.5)
    # + W_r x_dst: node0 (10, 100); node1 has no neighbours: mean 0 -> b + W_r*2 = (0.5+20, -0.5+200)
    assert torch.allclose(out, torch.tensor([[14.0, 108.5], [20.5, 199.5]]))


def test_segment_softmax_known_answer():
    src = torch.tensor([[0.0], [math.log(3.0)], [5.0]])
    idx = torch.tensor([0, 0, 1])
    out = R.segment_softmax(src, idx, 3)
    assert torch.allclose(out, torch.tensor([[0.25], [0.75], [1.0]]), atol=1e-7)


def test_gat_known_answer_single_head():
    # 2 nodes, edge 0->1 plus self loops; lin = identity on 1 feature, att_src = 1, att_dst = 2, bias 0.5
    conv = R.GATConv(1, 1, heads=1, concat=True, add_self_loops=True)
    with torch.no_grad():
        conv.lin_src.weight.fill_(1.0)
        conv.att_src.fill_(1.0)
        conv.att_dst.fill_(2.0)
        conv.bias.fill_(0.5)
    x = torch.tensor([[1.0], [-1.0]])
    out = conv(x, torch.tensor([[0], [1]]))
    # node0: only its loop: alpha = 1 -> out = x0 + 0.5 = 1.5
    # node1: edges from 0 and from itself: raw0 = a_s(0) + a_d(1) = 1 - 2 = -1 -> leaky = -0.2
    #                                      raw1 = -1 - 2 = -3 -> leaky = -0.6
    w0, w1 = math.exp(-0.2), math.exp(-0.6)
    exp1 = (w0 * 1.0 + w1 * -1.0) / (w0 + w1) + 0.5
    assert torch.allclose(out, torch.tensor([[1.5], [exp1]]), atol=1e-6)


def test_gat_removes_existing_self_loops_and_uses_fill_value():
    conv = R.GATConv((2, 2), 3, heads=2, concat=False, add_self_loops=True, edge_dim=3,
                     fill_value=torch.zeros(3, dtype=torch.float64))
    x = torch.randn(4, 2)
    ei = torch.tensor([[0, 1, 2, 2], [1, 1, 1, 3]])  # (1->1) is a genuine self loop
    ea = torch.randn(4, 3)
    out, (ei2, alpha) = conv(x, ei, ea, return_alpha=True)
    assert ei2.shape[1] == 3 + 4  # self loop removed, 4 loops appended after the real edges
    assert torch.equal(ei2[:, -4:], torch.arange(4).repeat(2, 1))
    # softmax rows sum to one per destination and head
    sums = R.scatter_sum(alpha, ei2[1], 4)
    assert torch.allclose(sums, torch.ones_like(sums), atol=1e-6)
    # the same-type conv ignores lin_dst even though it exists as a separate module
    conv.lin_dst.weight.data.add_(10.0)
    out2 = conv(x, ei, ea)
    assert torch.equal(out, out2)


def test_hetero_conv_sum_and_skip_rules():
    convs = {("a", "r1", "b"): R.SAGEConv((2, 3), 4), ("b", "r2", "b"): R.SAGEConv((3, 3), 4), ("a", "r3", "a"): R.SAGEConv((2, 2), 5)}
    hc = R.HeteroConv(convs, aggr="sum")
    x = {"a": torch.randn(3, 2), "b": torch.randn(2, 3)}
    ei = {("a", "r1", "b"): torch.tensor([[0, 2], [1, 1]]), ("b", "r2", "b"): torch.empty((2, 0), dtype=torch.int64)}
    out = hc(x, ei)
    assert set(out.keys()) == {"b"}  # ("a","r3","a") has no edge-level argument -> skipped
    exp = convs["a", "r1", "b"]((x["a"], x["b"]), ei["a", "r1", "b"]) + convs["b", "r2", "b"](x["b"], ei["b", "r2", "b"])
    assert torch.allclose(out["b"], exp)
    assert list(hc.state_dict().keys())[0].startswith("convs.a__r1__b.")


def test_leaf_pool_known_answer():
    x = torch.tensor([[2.0, 0.0], [4.0, 2.0], [9.0, 9.0]])
    ei = torch.tensor([[0, 1, 2], [0, 0, 1]])  # leaves 0,1 -> virtual 0; leaf 2 -> virtual 1
    out = R.LeafPool()(x, ei)
    assert out.shape == (3, 2)  # PyG sizes the output by x.size(0); the caller slices
    assert torch.allclose(out[:2], torch.tensor([[3.0, 1.0], [9.0, 9.0]]))
    assert torch.all(out[2] == 0)


def test_masked_cross_entropy_matches_manual():
    pred = torch.tensor([[2.0, 0.0, 0.0], [0.0, 1.0, 0.0], [5.0, 5.0, 5.0]])
    label = torch.tensor([0, 2, 1])
    mask = label != 1
    got = R.cross_entropy_loss(pred, label, mask)
    l0 = -math.log(math.exp(2) / (math.exp(2) + 2))
    l1 = -math.log(1 / (2 + math.e))
    assert abs(float(got) - (l0 + l1) / 2) < 1e-6


def test_parameter_counts_match_survey_appendix_c2():
    net = omodels.HeterogeneousNetwork({"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3)
    assert sum(p.numel() for p in net.parameters()) == 126568
    ht = omodels.HeterogeneousNeuralTreeNetwork(
        {"object": 306, "room": 6, "object-room": 6, "room-room": 6, "object_virtual": 306, "room_virtual": 6}, output_dim=26,
        conv_block="GraphSAGE", hidden_dim=128, num_layers=4, disable_initialization=True)
    assert sum(p.numel() for p in ht.parameters()) == 818180


def test_state_dict_keys_follow_pyg_naming():
    net = omodels.HeterogeneousNetwork({"objects": 303, "rooms": 3}, output_dim=26, conv_block="GAT_edge", GAT_hidden_dims=[8, 8],
                                       GAT_heads=[2, 2, 2], GAT_concats=[False, False, False])
    keys = set(net.state_dict().keys())
    base = "convs.0.convs.objects__objects_to_rooms__rooms."
    for k in ("att_src", "att_dst", "att_edge", "bias", "lin_src.weight", "lin_dst.weight", "lin_edge.weight"):
        assert base + k in keys
    # later layers are built from an int width: lin_dst IS lin_src (one tensor under two names)
    sd = net.state_dict()
    assert sd["convs.1.convs.objects__objects_to_rooms__rooms.lin_src.weight"].data_ptr() == \
        sd["convs.1.convs.objects__objects_to_rooms__rooms.lin_dst.weight"].data_ptr()


@pytest.mark.parametrize("block", ["GraphSAGE", "GAT", "GAT_edge"])
def test_float32_vs_float64_and_gradcheck(block):
    from hydra_gnn_amd import workloads

    torch.manual_seed(1)
    edge = block == "GAT_edge"
    dims = {"objects": 9, "rooms": 3} if edge else {"objects": 12, "rooms": 6}
    kw = dict(input_dim_dict=dims, output_dim=5, conv_block=block, hidden_dim=8, num_layers=2, GAT_hidden_dims=[4],
              GAT_heads=[2, 2], GAT_concats=[True, False], dropout=0.0)
    net = omodels.HeterogeneousNetwork(**kw).eval()
    batch = workloads.mp3d_like_batch(2, seed=3, relative_pos=edge, sem_dim=6)
    p32 = net(batch)
    n64 = omodels.HeterogeneousNetwork(**kw).double().eval()
    n64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    b64 = batch.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    for et in b64.edge_types:
        if "edge_attr" in b64[et]:
            b64[et].edge_attr = b64[et].edge_attr.double()
    p64 = n64(b64)
    assert torch.allclose(p32.double(), p64, atol=1e-5, rtol=1e-5)
    # gradcheck of the whole stack w.r.t. one weight matrix
    name, w = next((n, p) for n, p in n64.named_parameters() if p.dim() == 2 and p.numel() < 200)

    def f(wv):
        sd = dict(n64.named_parameters())
        old = sd[name].data
        sd[name].data = wv
        try:
            return torch.func.functional_call(n64, {name: wv}, (b64,))
        finally:
            sd[name].data = old

    assert torch.autograd.gradcheck(f, (w.detach().clone().requires_grad_(True),), eps=1e-6, atol=1e-5, rtol=1e-4)


# ---- GCN / GIN / BatchNorm -------------------------------------------------------------------------
def test_gcn_known_answer_and_self_loop_replacement():
    # edges j->i: 0->1, 1->0, 2->1.  With one loop per node: deg = (2, 3, 1); lin = identity, bias 0.5, x = (1, 2, 3)
    conv = R.GCNConv(1, 1)
    with torch.no_grad():
        conv.lin.weight.fill_(1.0)
        conv.bias.fill_(0.5)
    x = torch.tensor([[1.0], [2.0], [3.0]])
    ei = torch.tensor([[0, 1, 2], [1, 0, 1]])
    want = torch.tensor([[1 / 2 + 2 / math.sqrt(6)], [2 / 3 + 1 / math.sqrt(6) + 3 / math.sqrt(3)], [3.0]]) + 0.5
    assert torch.allclose(conv(x, ei), want, atol=1e-6)
    # an existing loop 2->2 is REPLACED by the added one (add_remaining_self_loops), not counted twice
    ei2 = torch.tensor([[0, 1, 2, 2], [1, 0, 1, 2]])
    assert torch.allclose(conv(x, ei2), want, atol=1e-6)


def test_gin_known_answer():
    mlp = torch.nn.Sequential(torch.nn.Linear(1, 1), torch.nn.ReLU(), torch.nn.Linear(1, 1))
    conv = R.GINConv(mlp, eps=0.5, train_eps=True)
    with torch.no_grad():
        mlp[0].weight.fill_(2.0), mlp[0].bias.fill_(1.0), mlp[2].weight.fill_(-1.0), mlp[2].bias.fill_(0.0)
    x = torch.tensor([[1.0], [2.0], [3.0]])
    ei = torch.tensor([[0, 2, 1], [1, 1, 0]])
    # s = 1.5 x_i + sum_j x_j = (3.5, 7, 4.5); 2 s + 1 = (8, 15, 10); relu; * -1
    assert torch.allclose(conv(x, ei), torch.tensor([[-8.0], [-15.0], [-10.0]]), atol=1e-6)
    assert list(dict(conv.named_parameters())) == ["eps", "nn.0.weight", "nn.0.bias", "nn.2.weight", "nn.2.bias"]


def test_gcn_gin_state_dict_keys_and_batchnorm_placement():
    kw = dict(input_dim=6, output_dim=15, hidden_dim=8, num_layers=3, dropout=0.0)
    gcn = omodels.HomogeneousNetwork(conv_block="GCN", **kw)
    assert sorted(gcn.state_dict()) == sorted(f"convs.{l}.{k}" for l in range(3) for k in ("bias", "lin.weight"))
    gin = omodels.HomogeneousNetwork(conv_block="GIN", **kw)
    keys = set(gin.state_dict())
    assert {"convs.0.eps", "convs.0.nn.0.weight", "convs.2.nn.2.bias", "batch_norms.0.module.weight", "batch_norms.2.module.running_var",
            "batch_norms.1.module.num_batches_tracked"} <= keys
    # homogeneous_network.py:133-134: BatchNorm sits between the conv and the relu of every layer but the last;
    # homogeneous_neural_tree_network.py:86-94: the H-tree loop never applies it
    g = torch.Generator().manual_seed(0)
    x = torch.randn(9, 6, generator=g)
    ei = torch.randint(0, 9, (2, 20), generator=g)
    data = type("D", (), dict(x=x, edge_index=ei, room_mask=torch.arange(9) < 3, init_edge_index=ei[:, :0], pool_edge_index=ei[:, :4]))
    gin.train()
    gin(data)
    assert [int(b.module.num_batches_tracked) for b in gin.batch_norms] == [1, 1, 0]
    tree = omodels.HomogeneousNeuralTreeNetwork(conv_block="GIN", disable_initialization=True, **kw)
    tree.train()
    tree(data)
    assert [int(b.module.num_batches_tracked) for b in tree.batch_norms] == [0, 0, 0]


@pytest.mark.parametrize("block", ["GCN", "GIN"])
def test_gcn_gin_gradcheck(block):
    torch.manual_seed(3)
    net = omodels.HomogeneousNetwork(input_dim=3, output_dim=4, conv_block=block, hidden_dim=5, num_layers=2, dropout=0.0).double()
    net.eval()  # BatchNorm on running statistics (gradcheck perturbs one input at a time)
    x = torch.randn(7, 3, dtype=torch.float64, requires_grad=True)
    ei = torch.randint(0, 7, (2, 15))
    mask = torch.arange(7) < 4

    def f(xx):
        return net(type("D", (), dict(x=xx, edge_index=ei, room_mask=mask)))

    assert torch.autograd.gradcheck(f, (x,), atol=1e-6)


def test_bf16_contract_restatement_equals_pyg_restatement_without_rounding():
    """oracle/bf16_emul.py follows the engine's project-then-aggregate algebra so that it can round where the engine rounds.
    With every rounding switched off it must be the SAME function as the PyG restatement (aggregate-then-project): logits,
    loss and every gradient agree to float64 round-off, dead last-layer convs get no gradient in both."""
    import copy

    from hydra_gnn_amd import workloads
    from oracle import bf16_emul, models as omodels

    torch.manual_seed(3)
    kw = dict(input_dim_dict={"objects": 32, "rooms": 32}, output_dim=26, conv_block="GraphSAGE", hidden_dim=32, num_layers=3, dropout=0.0)
    ora = omodels.HeterogeneousNetwork(**kw)
    g = workloads.big_hetero_graph(n_obj=300, n_rooms=12, deg=5, feat_dim=32, seed=4)
    o64 = copy.deepcopy(ora).double()
    b64 = g.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    y = g["rooms"].y
    pred = o64(b64)
    loss = o64.loss(pred, y, y != 25)
    loss.backward()
    logits, l2, grads = bf16_emul.sage_hetero_bf16(ora, g, dtype=torch.float64, rounding=False)
    # the aggregate-first evaluation of objects -> rooms (the engine's choice at >= 32 768 objects) is the same function too
    af = {(l, ("objects", "objects_to_rooms", "rooms")) for l in range(3)}
    la, l2a, ga = bf16_emul.sage_hetero_bf16(ora, g, dtype=torch.float64, rounding=False, agg_first=af)
    torch.testing.assert_close(la, logits, atol=1e-10, rtol=1e-10)
    torch.testing.assert_close(l2a, l2, atol=1e-10, rtol=1e-10)
    for name, gr in grads.items():
        assert (gr is None) == (ga[name] is None), name
        if gr is not None:
            torch.testing.assert_close(ga[name].double(), gr.double(), atol=1e-6, rtol=1e-6, msg=lambda m: f"agg-first {name}: {m}")
    # not 1e-12: the restatement sums the root weights / biases of a destination type in fp32 first (as the engine's pack does)
    torch.testing.assert_close(logits, pred.detach(), atol=5e-7, rtol=1e-6)
    torch.testing.assert_close(l2, loss.detach(), atol=5e-7, rtol=1e-6)
    ref = dict(o64.named_parameters())
    n_checked = 0
    for name, p in ref.items():
        if p.grad is None:
            assert name not in grads or grads[name] is None, name
            continue
        torch.testing.assert_close(grads[name].double(), p.grad, atol=5e-7, rtol=1e-6, msg=lambda m: f"{name}: {m}")
        n_checked += 1
    assert n_checked >= 20
    # and the rounding really rounds: bf16 contract differs from the exact function by about 2^-9 relative
    lr, _, _ = bf16_emul.sage_hetero_bf16(ora, g, dtype=torch.float64, rounding=True)
    rel = float((lr - logits).abs().max() / logits.abs().max())
    assert 1e-4 < rel < 5e-2, rel
