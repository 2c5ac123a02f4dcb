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


class NativeComm:
    """RCCL communicator owned by ``libhydra_mp.so`` (``hmp_comm_*``, csrc/comm.hip): the step's ONE all-reduce is enqueued
    on the stream the step's kernels run on, between the two native phases -- no second stream and no event hand-offs as in
    torch's ProcessGroupNCCL.  torch.distributed is only the rendezvous channel for the 128-byte unique id.

    ``NativeComm.from_process_group()`` needs an initialised default process group (any backend); a 1-rank communicator
    (``NativeComm.single()``) prices the collective path on one GPU."""

    def __init__(self, handle, rank: int, world: int):
        self._h, self.rank, self.world = handle, rank, world

    @classmethod
    def single(cls) -> "NativeComm":
        import ctypes as C

        from . import _lib

        lib = _lib.require_device()
        buf = (C.c_ubyte * 128)()
        _lib.check(lib.hmp_comm_unique_id(buf))
        h = C.c_void_p()
        _lib.check(lib.hmp_comm_create(buf, 0, 1, C.byref(h)))
        return cls(h, 0, 1)

    @classmethod
    def from_process_group(cls, device: torch.device, group=None) -> "NativeComm":
        import ctypes as C

        import torch.distributed as dist

        from . import _lib

        lib = _lib.require_device()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            _lib.check(lib.hmp_comm_unique_id(buf))
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.tensor(list(buf), dtype=torch.uint8, device=device if on_gpu else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ids = bytes(t.cpu().tolist())
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.hmp_comm_create(ids, rank, world, C.byref(h)))
        return cls(h, rank, world)

    def query(self):
        """(rank count, rank) as RCCL reports them for this communicator (ncclCommCount / ncclCommUserRank)."""
        import ctypes as C

        from . import _lib

        cnt, rk = C.c_int32(), C.c_int32()
        _lib.check(_lib.load().hmp_comm_query(self._h, C.byref(cnt), C.byref(rk)))
        return cnt.value, rk.value

    def all_reduce_sum_(self, buf: torch.Tensor, n: int) -> None:
        """sum ``buf[:n]`` (fp32, device) over the ranks, in place, on torch's CURRENT stream."""
        from . import _lib

        assert buf.dtype == torch.float32 and buf.is_cuda and buf.is_contiguous() and 0 <= n <= buf.numel()
        _lib.check(_lib.load().hmp_comm_allreduce_sum_f32(self._h, buf.data_ptr(), int(n), _lib.stream_ptr()))

    def broadcast_(self, buf: torch.Tensor, n: int, root: int = 0) -> None:
        from . import _lib

        assert buf.dtype == torch.float32 and buf.is_cuda and buf.is_contiguous() and 0 <= n <= buf.numel()
        _lib.check(_lib.load().hmp_comm_broadcast_f32(self._h, buf.data_ptr(), int(n), int(root), _lib.stream_ptr()))

    def close(self) -> None:
        if self._h is not None:
            from . import _lib

            _lib.load().hmp_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
