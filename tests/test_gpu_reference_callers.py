"""The two callers the engine must stay drop-in for, exercised the way the reference calls them:

* the loop body of ``BaseTrainingJob.train`` (``base_training_job.py:197-219``): ``opt.zero_grad(); pred = net(batch.to(device));
  loss = net.loss(pred, label, mask); loss.backward(); opt.step()`` with ``torch.optim.Adam(net.parameters(), lr, weight_decay)``
  (:181-185) -- torch's optimiser updating the engine's parameter views in place;
* ``GnnModel.__init__`` / ``infer`` of ``bin/room_classification_server`` (:203-204, :285-286): ``load_state_dict(torch.load(
  "model_weights.pth"))`` (strict) and ``model(data.to(device)).argmax(dim=1).cpu()`` under ``no_grad`` on ONE scene graph.

Both are compared with the oracle models driven by the same code.
"""
import copy
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import workloads  # noqa: E402
from hydra_gnn_amd.data import collate  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork, HeterogeneousNeuralTreeNetwork  # noqa: E402
from oracle import models as omodels  # noqa: E402

DEV = "cuda:0"
KW = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3, dropout=0.0)
HT_DIMS = {"object": 306, "room": 6, "object-room": 6, "room-room": 6, "object_virtual": 306, "room_virtual": 6}


def test_reference_training_loop_with_torch_adam_matches_oracle():
    torch.manual_seed(0)
    ora = omodels.HeterogeneousNetwork(**KW)
    net = HeterogeneousNetwork(**KW)
    net.load_state_dict(ora.state_dict(), strict=True)
    net = net.to(DEV)
    o64 = copy.deepcopy(ora).double()
    opt = torch.optim.Adam(net.parameters(), lr=0.002, weight_decay=0.001)
    opt_ref = torch.optim.Adam(o64.parameters(), lr=0.002, weight_decay=0.001)
    rng = np.random.default_rng(3)
    graphs = [workloads.mp3d_like_graph(rng) for _ in range(12)]
    losses, losses_ref = [], []
    for it in range(6):  # a different batch every step, as the DataLoader delivers
        batch = collate([graphs[i] for i in rng.choice(12, size=4, replace=False)])
        y = batch["rooms"].y
        # reference loop body
        opt.zero_grad()
        gb = batch.to(DEV)
        pred = net(gb)
        label = gb["rooms"].y
        mask = label != 25
        loss = net.loss(pred, label, mask)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        # oracle
        b64 = batch.to("cpu")
        for t in b64.node_types:
            b64[t].x = b64[t].x.double()
        opt_ref.zero_grad()
        loss_ref = o64.loss(o64(b64), y, y != 25)
        loss_ref.backward()
        opt_ref.step()
        losses_ref.append(loss_ref.item())
    np.testing.assert_allclose(losses, losses_ref, rtol=5e-5, atol=5e-5)
    ref = dict(o64.named_parameters())
    for name, p in net.named_parameters():
        if ref[name].grad is None:  # last-layer convs into `objects` never reach the loss: torch.optim.Adam skips them in both
            assert torch.equal(p.detach().cpu(), dict(ora.named_parameters())[name].detach()), name
            continue
        d = (p.detach().cpu().double() - ref[name].detach()).abs()
        assert float((d > 5e-5).double().mean()) < 0.02, name  # Adam is ill-conditioned where |g| ~ eps (see test_gpu_models.py)
        assert float(d.max()) <= 6 * 0.002 * 2.1, name


@pytest.mark.parametrize("htree", [False, True])
def test_server_style_inference_on_one_scene_graph(tmp_path, htree):
    torch.manual_seed(1)
    if htree:
        kw = dict(input_dim_dict=HT_DIMS, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3,
                  disable_initialization=True, dropout=0.25)
        ora = omodels.HeterogeneousNeuralTreeNetwork(**kw)
        model = HeterogeneousNeuralTreeNetwork(**kw)
        data = workloads.htree_batch(1, seed=77)
        readout = "room_virtual"
    else:
        kw = dict(KW, dropout=0.25)
        ora = omodels.HeterogeneousNetwork(**kw)
        model = HeterogeneousNetwork(**kw)
        data = collate([workloads.mp3d_like_graph(np.random.default_rng(5))])
        readout = "rooms"
    # GnnModel.__init__: weights come from model_weights.pth, strict load (the file is one we wrote: weights_only load)
    weight_path = tmp_path / "model_weights.pth"
    torch.save(ora.state_dict(), weight_path)
    model.load_state_dict(torch.load(weight_path, weights_only=True))
    model.to(DEV)
    model.eval()
    ora.eval()
    # GnnModel.infer
    with torch.no_grad():
        pred = model(data.to(DEV)).argmax(dim=1).cpu()
        ref_logits = ora(data)
    assert pred.shape[0] == data[readout].num_nodes
    ref = ref_logits.argmax(dim=1)
    # argmax may legitimately differ only where the two best logits tie within the 1e-5 parity tolerance
    top2 = ref_logits.topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(pred[decided], ref[decided])
    # latency of the server's inference call (forward + argmax + D2H), one graph at a time
    gdata = data.to(DEV)
    for _ in range(5):
        with torch.no_grad():
            model(gdata).argmax(dim=1).cpu()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        with torch.no_grad():
            model(gdata).argmax(dim=1).cpu()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(10):
        with torch.no_grad():
            ora(data).argmax(dim=1)
    dt_ref = (time.perf_counter() - t0) / 10
    print(f"server-style inference ({'H-tree' if htree else 'hetero'}), 1 graph: engine {1e3 * dt:.3f} ms, oracle on CPU {1e3 * dt_ref:.3f} ms")
    assert dt < 5e-3
