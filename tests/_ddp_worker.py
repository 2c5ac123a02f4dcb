"""Child process of tests/test_gpu_multirank.py: one data-parallel rank of the native training step.  Ranks SHARE cuda:0 and
reduce through gloo (RCCL refuses two ranks on one device); the engine-side protocol -- phase A, ONE flat all-reduce of
[gradient sums | loss_sum | count], phase B, initial parameter broadcast -- is the one the RCCL path runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hydra-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, out_path, n_graphs, steps, use_graph = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4],
                                                               int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hydra_gnn_amd import parallel, workloads
    from hydra_gnn_amd.data import collate
    from hydra_gnn_amd.models import HeterogeneousNetwork

    torch.cuda.set_device(0)
    torch.manual_seed(100 + rank)  # ranks start different: the step object broadcasts rank 0's weights
    net = HeterogeneousNetwork({"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=64, num_layers=3,
                               dropout=0.0).to("cuda:0")
    rng = np.random.Generator(np.random.PCG64(99))
    graphs = [workloads.mp3d_like_graph(rng) for _ in range(n_graphs)]
    graphs[0]["rooms"].y[:] = 25  # unequal valid-label counts across ranks
    mine = parallel.shard_graphs(n_graphs, rank, world)
    batch = collate([graphs[i] for i in mine]).to("cuda:0")
    step = net.train_step(lr=0.002, weight_decay=0.001, ignored_label=25, use_graph=bool(use_graph), process_group=True)
    losses = []
    for _ in range(steps):
        step(batch, batch["rooms"].y)
        losses.append(step.loss())
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()
    torch.save({"params": flat, "losses": losses, "steps": step.steps_taken()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
