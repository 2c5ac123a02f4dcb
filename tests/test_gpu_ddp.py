"""Exercises the data-parallel launch structure on ONE GPU: a 1-rank RCCL ("nccl") process group, two captured hipGraphs
per step with a real all-reduce of the flat gradient buffer between them.  The result must equal the plain single-GPU
fused step bit for bit (a 1-rank sum is the identity).  Multi-rank numerics are covered by tests/test_parallel_gloo.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from hydra_gnn_amd import workloads  # noqa: E402
from hydra_gnn_amd.models import HeterogeneousNetwork  # noqa: E402

DEV = "cuda:0"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_group():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def run(steps, **kw):
    torch.manual_seed(0)
    net = HeterogeneousNetwork({"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3,
                               dropout=0.25).to(DEV)
    batch = workloads.config2_batch(8).to(DEV)
    y = batch["rooms"].y
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, seed=11, **kw)
    losses = []
    for _ in range(steps):
        step(batch, y)
        losses.append(step.loss())
    torch.cuda.synchronize()
    return losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_phase_step_with_rccl_allreduce_equals_single_phase(nccl_group, use_graph):
    ref_losses, ref_params = run(6, use_graph=use_graph)
    losses, params = run(6, use_graph=use_graph, process_group=True, force_collective=True)
    assert losses == ref_losses
    assert torch.equal(params, ref_params)
