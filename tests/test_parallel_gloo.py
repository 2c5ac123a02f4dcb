"""world_size-2 gloo test (CPU) of the data-parallel protocol: graph sharding + ONE flat all-reduce of
[gradient sums | loss_sum | count] reproduces the single-process full-batch gradient / loss exactly as the
count-weighted mean, also when the ranks hold different numbers of valid labels.

The per-rank compute is done by the oracle here (no GPU in this container); the protocol code under test is
hydra_gnn_amd.parallel, which engine.TrainStep drives on the GPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hydra_gnn_amd import parallel, workloads
from hydra_gnn_amd.data import collate
from oracle import models as omodels

KW = dict(input_dim_dict={"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=16, num_layers=3, dropout=0.0)
N_GRAPHS = 5


def make_graphs():
    rng = np.random.Generator(np.random.PCG64(99))
    gs = [workloads.mp3d_like_graph(rng) for _ in range(N_GRAPHS)]
    gs[0]["rooms"].y[:] = 25  # rank 0's first graph has no valid label at all: unequal counts across ranks
    return gs


def flat_sum_grads(net, batch):
    """[sum-gradient of every parameter that gets one | loss_sum | count] as the engine lays it out."""
    y = batch["rooms"].y
    mask = y != 25
    pred = net(batch)
    loss_sum = torch.nn.functional.cross_entropy(pred[mask], y[mask], reduction="sum") if bool(mask.any()) else pred.sum() * 0
    net.zero_grad()
    loss_sum.backward()
    gs = [p.grad.reshape(-1) for p in net.parameters() if p.grad is not None]
    flat = torch.cat(gs + [loss_sum.detach().reshape(1), mask.sum().reshape(1).to(torch.float32)])
    return flat, flat.numel() - 2


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = omodels.HeterogeneousNetwork(**KW)
    flat_w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    if rank == 1:
        flat_w += 1.0  # ranks start different on purpose ...
    parallel.broadcast_parameters(flat_w)  # ... and the initial broadcast makes them identical
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(flat_w[off:off + p.numel()].view_as(p))
            off += p.numel()
    graphs = make_graphs()
    mine = parallel.shard_graphs(len(graphs), rank, world)
    buf, n_active = flat_sum_grads(net, collate([graphs[i] for i in mine]))
    parallel.allreduce_flat(buf, n_active)
    g, loss, count = parallel.finish_gradients(buf, n_active)
    out[rank] = (g.clone(), float(loss), float(count), mine)
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_flat_allreduce_equals_full_batch():
    torch.manual_seed(0)
    net = omodels.HeterogeneousNetwork(**KW)
    graphs = make_graphs()
    full = collate(graphs)
    y = full["rooms"].y
    mask = y != 25
    net.zero_grad()
    loss = net.loss(net(full), y, mask)
    loss.backward()
    g_full = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None])

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(2, free_port(), out), nprocs=2, join=True)
    (g0, l0, c0, m0), (g1, l1, c1, m1) = out[0], out[1]
    assert sorted(m0 + m1) == list(range(N_GRAPHS)) and not set(m0) & set(m1)
    assert torch.equal(g0, g1) and l0 == l1 and c0 == c1 == float(mask.sum())
    torch.testing.assert_close(g0, g_full, atol=1e-6, rtol=1e-5)
    assert abs(l0 - float(loss)) < 1e-5


def test_sharding_functions():
    assert parallel.shard_graphs(8, 3, 8) == [3]                       # one graph per rank at B == world
    assert parallel.shard_graphs(128, 0, 8) == list(range(0, 128, 8))  # config 4: 16 graphs per rank
    parts = [parallel.shard_graphs_balanced([5, 1, 9, 3, 3, 7], r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([5, 1, 9, 3, 3, 7][i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 3
