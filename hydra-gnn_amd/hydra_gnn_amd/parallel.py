"""Data-parallel host logic (one process per GPU, ``torch.distributed``; backend "nccl" is RCCL on ROCm).

The reference has no multi-GPU code at all: a batch is a disjoint union of scene graphs
(``base_training_job.py:164-168`` / PyG collate), so graphs shard across ranks with NO data-path exchange; the only
collective is ONE all-reduce(sum) per step over a single flat fp32 buffer

    [ gradient SUMS of the active parameters (n_active) | loss_sum | valid-label count ]

after which every rank scales by 1/count and applies the identical Adam update.  Summing and dividing by the GLOBAL
count (instead of averaging per-rank means) makes the N-rank step equal to the single-process step on the whole batch
even when ranks hold different numbers of valid labels.

Gradient buffers here are 0.5-13 MB: latency-bound on xGMI, so the buffer is never bucketed.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def shard_graphs(n_graphs: int, rank: int, world: int) -> List[int]:
    """Graph ids of ``rank``: round-robin ``{i : i mod world == rank}`` (one-graph-per-rank at B == world)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_graphs, world))


def shard_graphs_balanced(edge_counts: Sequence[int], rank: int, world: int) -> List[int]:
    """Alternative split: longest-processing-time greedy on per-graph edge counts (deterministic on every rank)."""
    order = sorted(range(len(edge_counts)), key=lambda i: (-int(edge_counts[i]), i))
    load = [0] * world
    mine: List[int] = []
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += int(edge_counts[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def allreduce_flat(buf: torch.Tensor, n_active: int, group=None, force: bool = False) -> torch.Tensor:
    """Sum ``buf[0 : n_active + 2]`` over the ranks in place (one collective) and return that slice.
    ``force`` issues the collective even in a 1-rank group (exercises the RCCL path on a single GPU)."""
    import torch.distributed as dist

    view = buf[: n_active + 2]
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
    return view


def finish_gradients(buf: torch.Tensor, n_active: int):
    """(mean gradient, mean loss, count) from a reduced flat buffer -- what the native Adam phase computes on device."""
    count = buf[n_active + 1].clamp(min=1.0)
    return buf[:n_active] / count, buf[n_active] / count, buf[n_active + 1]


def broadcast_parameters(flat: torch.Tensor, group=None, src: int = 0) -> None:
    """One initial broadcast so every rank starts from rank 0's weights."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
