"""BASELINE config 5 ("synthetic hetero scene graph ~1M object + 10k room nodes, avg deg 16, hidden=256 bf16") under an oracle.

(a) at 4 x 10^4 objects -- a size the float64 oracle holds -- with the decisions of the 10^6-object regime pinned
    (HMP_BF16_ALL=1: every GEMM call takes the bf16 kernel, as at full size):
      * fp32 mode against the PyG restatement in float64 (oracle/models.py): logits, loss, ALL gradients;
      * bf16 mode against oracle/bf16_emul.py = the same model with a round-to-bf16 exactly where the engine rounds (Z / G / dZ /
        H storage, GEMM operands).  Tolerance DERIVED, not borrowed from the engine's own fp32 mode: the restatement is evaluated
        twice, accumulating in float64 and in float32 (a different summation order than the engine's); where the two differ a
        value sat within accumulation noise of a bf16 rounding boundary and flipped.  The engine must sit inside a small multiple
        of that spread -- and far inside the distance between the bf16 contract and the exact function (the test has power).
(b) at full size (10^6 objects, 16 M object-object edges): properties that need no oracle -- status word 0, the plan bit-exact
    against torch.sort(stable=True), two runs bit-identical, bf16-mode loss within the bf16-vs-exact distance measured in (a).
"""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import _lib, ops, workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402
from oracle import bf16_emul  # noqa: E402
from oracle import models as omodels  # noqa: E402

DEV = "cuda:0"
KW = dict(input_dim_dict={"objects": 256, "rooms": 256}, output_dim=26, conv_block="GraphSAGE", hidden_dim=256, num_layers=3, dropout=0.25)
N_OBJ, N_ROOMS = 40000, 400
NT = {"objects": 0, "rooms": 1}


def build(seed=0):
    torch.manual_seed(seed)
    ora = omodels.HeterogeneousNetwork(**KW)
    net = HeterogeneousNetwork(**KW)
    net.load_state_dict(ora.state_dict(), strict=True)
    return ora, net.to(DEV)


def make_replay(net):
    """oracle dropout_fn that replays the keep-masks the engine drew in its LAST training forward (hmp_dropout_mask)"""
    lib = _lib.require_device()

    def replay(x, pp, training, tag):
        if not training or pp == 0:
            return x
        layer, t = tag[1:].split(".", 1)
        n, F = x.shape
        m = torch.zeros(n * F, dtype=torch.uint8, device=DEV)
        _lib.check(lib.hmp_dropout_mask(net._seed, net._rng_step, int(layer) * 8 + NT[t], pp, n, F, m.data_ptr(), _lib.stream_ptr()))
        keep = m.view(n, F).cpu().to(x.dtype)
        return x * keep / (1.0 - pp)

    return replay


def engine_fwd_bwd(net, g):
    for p in net.parameters():
        p.grad = None
    net.train()
    pred = net(g)
    y = g["rooms"].y
    loss = net.loss(pred, y, y != 25)
    loss.backward()
    grads = {k: (p.grad.detach().cpu().clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    return pred.detach().cpu().clone(), float(loss), grads


def nerr(a, b):
    """max |a - b| relative to the tensor's scale max |b|"""
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))


@pytest.fixture(scope="module")
def graph():
    return workloads.big_hetero_graph(n_obj=N_OBJ, n_rooms=N_ROOMS, seed=31)


def keep_mask(net, l, t, n, F, p):
    lib = _lib.require_device()
    m = torch.zeros(n * F, dtype=torch.uint8, device=DEV)
    _lib.check(lib.hmp_dropout_mask(net._seed, net._rng_step, l * 8 + NT[t], p, n, F, m.data_ptr(), _lib.stream_ptr()))
    return m.view(n, F).cpu().bool()


def test_config5_fp32_mode_matches_the_float64_oracle(graph):
    """training mode, dropout 0.25 with replayed masks, hidden 256, 3 layers, 40 400 nodes / 720 000 edges (the stand-alone
    launch sequence: grouped fp32 MFMA GEMMs, one-wave-per-row aggregation): logits + loss + every gradient at 1e-5.

    ReLU at this size: of 2 x 10^7 hidden pre-activations a few lie within fp32 round-off of zero, and the engine (fp32) and the
    oracle (float64) then take different sides; ONE such unit moves a weight-gradient element by a whole node's contribution
    (~1e-3 of the tensor's scale, measured).  That is a property of ReLU'(0), not an engine error, so for units with
    |pre-activation| < 1e-5 -- and only for those -- the oracle takes the side the engine took (read back through
    hmp_net_hidden); how many disagreed is printed.  Everything else is compared at the north_star tolerance."""
    ora, net = build()
    g = graph.to(DEV)
    pred, loss, grads = engine_fwd_bwd(net, g)
    assert net.native().read_state()[1] == 0
    p = KW["dropout"]
    eng_h = {(l, t): net.native().hidden(l + 1, t).cpu() for l in range(2) for t in ("objects", "rooms")}
    o64 = copy.deepcopy(ora).double()
    o64.train()
    stats = {"ambiguous": 0, "flipped": 0}

    def act_drop(x_dict, l):
        out = {}
        for t, pre in x_dict.items():
            keep = keep_mask(net, l, t, pre.size(0), pre.size(1), p)
            pos = pre > 0
            amb = (pre.detach().abs() < 1e-5) & keep
            eng_pos = eng_h[(l, t)] > 0
            stats["ambiguous"] += int(amb.sum())
            stats["flipped"] += int((amb & (eng_pos != pos)).sum())
            m = torch.where(amb, eng_pos, pos) & keep
            # the engine's dropout keep bits must be the replayed ones wherever the unit is alive
            assert bool((torch.signbit(eng_h[(l, t)]) == ~keep).all()), "keep-mask replay does not match the engine's sign bits"
            out[t] = pre * m.to(pre.dtype) / (1.0 - p)
        return out

    o64._act_drop = act_drop
    b64 = graph.to("cpu")
    for t in b64.node_types:
        b64[t].x = b64[t].x.double()
    y = graph["rooms"].y
    pred_ref = o64(b64)
    loss_ref = o64.loss(pred_ref, y, y != 25)
    loss_ref.backward()
    torch.testing.assert_close(pred.double(), pred_ref.detach(), atol=1e-5, rtol=1e-5)
    assert abs(loss - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref)))
    worst = 0.0
    for name, q in o64.named_parameters():
        if q.grad is None:
            assert grads[name] is None, name
            continue
        assert grads[name] is not None, name
        # a weight / bias gradient element is a sum over up to 40 000 nodes x 16 edges of fp32 products of either sign: fp32
        # accumulation alone leaves ~sqrt(40 000) x 2^-24 = 1.2e-5 of the tensor's scale (measured worst 1.1e-5 .. 1.7e-5 over
        # dropout masks): compared at 3e-5 of the scale plus 1e-5 relative per element
        scale = float(q.grad.abs().max())
        torch.testing.assert_close(grads[name].double(), q.grad, atol=3e-5 * scale, rtol=1e-5, msg=lambda m: f"{name}: {m}")
        worst = max(worst, nerr(grads[name], q.grad))
    print(f"config-5 shape, fp32 mode vs float64 oracle: logits {nerr(pred, pred_ref.detach()):.2e}, worst gradient {worst:.2e} of "
          f"its scale; ReLU units within 1e-5 of zero: {stats['ambiguous']}, of which the two sides disagreed on {stats['flipped']}")


def test_config5_bf16_mode_matches_the_bf16_contract_oracle(graph, monkeypatch):
    monkeypatch.setenv("HMP_BF16_ALL", "1")  # the 10^6-object regime's decisions at a size the oracle can hold
    monkeypatch.setenv("HMP_FUSE", "0")      # ... incl. its launch sequence (stand-alone GEMMs: the ones that run in bf16)
    ora, net = build(seed=1)
    net.native().set_compute("bf16")
    g = graph.to(DEV)
    pred, loss, grads = engine_fwd_bwd(net, g)
    assert net.native().read_state()[1] == 0
    replay = make_replay(net)
    # the convs the engine evaluated aggregate-first (objects -> rooms at this size): the contract rounds where the engine rounds
    nat = net.native()
    af = {(l, tuple(nat.layers[l].convs[c].edge_type)) for (l, c), on in nat._agg_first.items() if on}
    assert af == {(l, ("objects", "objects_to_rooms", "rooms")) for l in range(3)}, af
    # the bf16 contract, accumulated in float64 and in float32, and the exact function (no rounding)
    l64, s64, g64 = bf16_emul.sage_hetero_bf16(ora, graph, dtype=torch.float64, rounding=True, dropout_fn=replay, training=True, agg_first=af)
    l32, s32, g32 = bf16_emul.sage_hetero_bf16(ora, graph, dtype=torch.float32, rounding=True, dropout_fn=replay, training=True, agg_first=af)
    lex, sex, gex = bf16_emul.sage_hetero_bf16(ora, graph, dtype=torch.float64, rounding=False, dropout_fn=replay, training=True)
    report = []

    def check(what, eng, a64, a32, aex):
        spread = nerr(a32, a64)          # accumulation-order noise amplified by bf16 rounding flips
        dist = nerr(aex, a64)            # what the bf16 contract costs against the exact function
        err = nerr(eng, a64)
        report.append((what, err, spread, dist))
        # the engine is a third accumulation order of the same contract: inside 6x the spread of the two CPU evaluations (the
        # spread is a maximum over a handful of discrete flips, itself noisy: measured ratios 0.6 .. 2.3) + 2e-5 of the tensor's
        # scale for plain fp32 summation error ...
        assert err <= 6.0 * spread + 2e-5, (what, err, spread)
        return err, dist

    e, d = check("logits", pred, l64, l32, lex)
    assert e < d / 2, ("logits: the test cannot tell the bf16 contract from the exact function", e, d)
    assert abs(loss - float(s64)) <= 6.0 * abs(float(s32) - float(s64)) + 2e-5 * abs(float(s64))
    n = 0
    tight = 0
    for name in g64:
        if g64[name] is None:
            assert grads[name] is None, name
            continue
        e, d = check(name, grads[name], g64[name], g32[name], gex[name])
        n += 1
        tight += int(e < d / 2)
    assert n == 30  # (4 + 4 + 2) live convs x (lin_l.weight, lin_l.bias, lin_r.weight)
    # ... and for (nearly) every tensor far inside the distance to the exact function: the comparison has power
    assert tight >= n - 2, report
    for r in report:
        print("bf16 contract: %-60s engine-vs-contract %.2e   f32-vs-f64 spread %.2e   contract-vs-exact %.2e" % r)
    # dead convs get no gradient
    assert all(grads[k] is None for k in grads if k.startswith("convs.2.") and k.split("__")[-1].startswith("objects"))


def test_config5_full_size_properties():
    """10^6 objects, 10^4 rooms, 16 M + 2 M + 60 k edges, hidden 256, bf16 mode: the workload of `bench.py --config 5`."""
    n_obj, n_rooms = 1_000_000, 10_000
    g = workloads.big_hetero_graph(n_obj=n_obj, n_rooms=n_rooms)  # BASE_SEED + 5: the bench's rank-0 graph
    gd = g.to(DEV)
    # ---- plan: bit-exact against a stable sort on the 16 M-edge list ------------------------------------------------
    ei = gd["objects", "objects_to_objects", "objects"].edge_index
    E = ei.size(1)
    plan = ops.GraphPlan(ei, n_obj)
    torch.cuda.synchronize()
    assert int(plan.status.item()) == 0
    rowptr, col, eid, t_rowptr, t_col, t_pos = plan._t
    order = torch.sort(ei[1], stable=True).indices
    assert torch.equal(eid[:E].long(), order)
    assert torch.equal(col[:E].long(), ei[0][order])
    rp = torch.zeros(n_obj + 1, dtype=torch.int64, device=DEV)
    rp[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n_obj), 0)
    assert torch.equal(rowptr[: n_obj + 1].long(), rp)
    t_order = torch.sort(ei[0], stable=True).indices
    assert torch.equal(t_col[:E].long(), ei[1][t_order])
    trp = torch.zeros(n_obj + 1, dtype=torch.int64, device=DEV)
    trp[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n_obj), 0)
    assert torch.equal(t_rowptr[: n_obj + 1].long(), trp)
    del plan, order, t_order, rp, trp
    # ---- two training steps from the same weights, twice: bit-identical, status 0 ---------------------------------
    y = gd["rooms"].y

    def run(precision, steps=2):
        torch.manual_seed(7)
        net = HeterogeneousNetwork(**KW).to(DEV)
        net.train()
        net.native().set_compute(precision)
        step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=20250225, use_graph=False)
        losses = []
        for _ in range(steps):
            step(gd, y)
            losses.append(step.loss())
        st, status = net.native().read_state()
        assert status == 0 and st == steps
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
        del step, net
        torch.cuda.empty_cache()
        return losses, flat

    la, pa = run("bf16")
    lb, pb = run("bf16")
    assert la == lb and torch.equal(pa, pb)
    assert all(np.isfinite(la)) and la[1] < la[0] + 0.5
    # ---- bf16 contract against the fp32 mode at full size: within the contract-vs-exact distance measured in (a)
    # (logits 3e-3 of scale at 4 x 10^4 objects; the loss is a mean over 10^4 rooms) ----------------------------------
    lf, pf = run("fp32", steps=1)
    assert abs(la[0] - lf[0]) <= 1e-2 * abs(lf[0]), (la[0], lf[0])
    print(f"config 5 full size: loss bf16 {la[0]:.6f} / fp32 {lf[0]:.6f} (rel {abs(la[0] - lf[0]) / abs(lf[0]):.2e})")
