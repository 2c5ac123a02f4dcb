"""Differentiable operators over the op-level C ABI (``include/hydra_mp.h`` sections 1-5 and 9).

The SAGE / GAT families run as ONE native program per step (:mod:`hydra_gnn_amd.engine`).  The homogeneous GCN / GIN
family of the reference's Stanford3DSG configurations (``models/utils.py:15-26``, ``models/homogeneous_network.py:93-97``;
SURVEY.md 8(f) row 2) works on graphs of 2..27 nodes and is composed here op by op: every function below is a
``torch.autograd.Function`` whose forward and backward are calls into ``libhydra_mp.so`` -- torch provides tensors, the
stream and the autograd tape only.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _dev(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.HydraMPError(f"{what} is on the CPU; hydra_gnn_amd has no CPU fallback")
    if t.dtype != torch.float32:
        raise _lib.HydraMPError(f"{what}: the native operators compute in fp32 (got {t.dtype})")


def _rows(t: torch.Tensor) -> torch.Tensor:
    """row-major with unit column stride (the kernels take a leading dimension)"""
    return t if (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.size(1)) else t.contiguous()


class GraphPlan:
    """CSR (by destination) + CSC lists of one square ``edge_index`` (``hmp_plan_build``), built once per batch and
    shared by every layer; ``dinv`` is GCN's ``deg^-1/2`` with the replaced self loops counted (``hmp_gcn_norm``)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, num_src: Optional[int] = None):
        """``num_src``: rows of the source side when the edge type is bipartite (LeafPool over ``room -> room_virtual``);
        the GCN / GIN operators need the square form"""
        if not edge_index.is_cuda:
            raise _lib.HydraMPError("edge_index is on the CPU; hydra_gnn_amd has no CPU fallback")
        lib = _lib.require_device()
        ei = edge_index.to(torch.int64).contiguous()
        E, d, n = int(ei.size(1)), ei.device, int(num_nodes)
        ns = n if num_src is None else int(num_src)
        self.n, self.n_src, self.E, self.device = n, ns, E, d
        i32 = lambda k: torch.empty(max(k, 1), dtype=torch.int32, device=d)
        self._t = [i32(n + 1), i32(E), i32(E), i32(ns + 1), i32(E), i32(E)]
        self.plan = _lib.Plan(ns, n, E, *[t.data_ptr() for t in self._t])
        scratch = torch.empty(max(int(lib.hmp_plan_scratch_bytes(E, ns, n)), 16), dtype=torch.uint8, device=d)
        self.status = torch.zeros(1, dtype=torch.int32, device=d)
        with torch.cuda.device(d):
            _lib.check(lib.hmp_plan_build(ei.data_ptr() if E > 0 else None, self.plan, scratch.data_ptr(), self.status.data_ptr(),
                                          _lib.stream_ptr()))
        self._ei, self._scratch = ei, scratch  # alive until the stream has consumed them
        self._dinv: Optional[torch.Tensor] = None

    @property
    def dinv(self) -> torch.Tensor:
        if self._dinv is None:
            self._dinv = torch.empty(max(self.n, 1), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().hmp_gcn_norm(self.plan, self._dinv.data_ptr(), _lib.stream_ptr()))
        return self._dinv


def _gemm(a, trans_a, b, trans_b, M, N, K) -> torch.Tensor:
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if M == 0 or N == 0:
        return c
    if K == 0:
        return c.zero_()
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().hmp_gemm_f32(a.data_ptr(), a.stride(0), trans_a, b.data_ptr(), b.stride(0), trans_b, c.data_ptr(),
                                            c.stride(0), M, N, K, _lib.stream_ptr()))
    return c


def _colsum(g: torch.Tensor) -> torch.Tensor:
    out = torch.empty(g.size(1), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(_lib.load().hmp_colsum(g.data_ptr(), g.stride(0), g.size(0), g.size(1), out.data_ptr(), _lib.stream_ptr()))
    return out


class _Project(torch.autograd.Function):
    """``x @ weight.T`` (nn.Linear layout) on the fp32 matrix pipe; no bias (biases ride in :func:`bias_act_drop`)."""

    @staticmethod
    def forward(ctx, x, weight):
        _lib.require_device()
        _dev(x, "x"), _dev(weight, "weight")
        x, weight = _rows(x), _rows(weight)
        ctx.save_for_backward(x, weight)
        return _gemm(x, 0, weight, 1, x.size(0), weight.size(0), x.size(1))

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _rows(g)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _gemm(g, 0, weight, 0, g.size(0), weight.size(1), weight.size(0))
        if ctx.needs_input_grad[1]:
            gw = _gemm(g, 1, x, 0, weight.size(0), weight.size(1), g.size(0))
        return gx, gw


class _WSum(torch.autograd.Function):
    """``hmp_segment_wsum``: GCN's symmetric-normalised sum (``w = plan.dinv``) or GIN's ``sum + (1 + eps) * self``."""

    @staticmethod
    def forward(ctx, z, plan: GraphPlan, gcn: bool, eps):
        _dev(z, "z")
        z = _rows(z)
        if z.size(0) != plan.n or plan.n_src != plan.n:
            raise _lib.HydraMPError(f"segment_wsum: {z.size(0)} rows for a plan over {plan.n_src} -> {plan.n} nodes (square plans only)")
        ctx.plan, ctx.gcn = plan, gcn
        ctx.save_for_backward(z, eps if eps is not None else z.new_empty(0))
        ctx.has_eps = eps is not None
        return _WSum._run(z, plan, gcn, eps, 0)

    @staticmethod
    def _run(z, plan, gcn, eps, transpose):
        out = torch.empty((z.size(0), z.size(1)), dtype=torch.float32, device=z.device)
        if z.numel() == 0:
            return out
        w = plan.dinv.data_ptr() if gcn else None
        with torch.cuda.device(z.device):
            _lib.check(_lib.load().hmp_segment_wsum(z.data_ptr(), z.stride(0), z.size(1), plan.plan, transpose, w,
                                                    eps.data_ptr() if eps is not None else None, out.data_ptr(), out.stride(0),
                                                    _lib.stream_ptr()))
        return out

    @staticmethod
    def backward(ctx, g):
        z, eps = ctx.saved_tensors
        eps = eps if ctx.has_eps else None
        g = _rows(g)
        gz = _WSum._run(g, ctx.plan, ctx.gcn, eps, 1) if ctx.needs_input_grad[0] else None
        geps = None
        if ctx.has_eps and ctx.needs_input_grad[3]:
            geps = torch.empty_like(eps)
            if z.numel() == 0:
                geps.zero_()
            else:
                with torch.cuda.device(z.device):
                    _lib.check(_lib.load().hmp_rowdot_sum(g.data_ptr(), g.stride(0), z.data_ptr(), z.stride(0), z.size(0), z.size(1),
                                                          geps.data_ptr(), _lib.stream_ptr()))
        return gz, None, None, geps


class _BiasActDrop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, act: int, p: float, seed: int, rng_step: int, rng_stream: int):
        _dev(x, "x")
        x = _rows(x)
        y = torch.empty((x.size(0), x.size(1)), dtype=torch.float32, device=x.device)
        if x.numel():
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().hmp_bias_act_drop_fwd(x.data_ptr(), x.stride(0), x.size(0), x.size(1),
                                                             bias.data_ptr() if bias is not None else None, int(act), float(p), seed,
                                                             rng_step, rng_stream, y.data_ptr(), y.stride(0), _lib.stream_ptr()))
        ctx.act, ctx.p, ctx.has_bias = int(act), float(p), bias is not None
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _rows(g)
        if ctx.act != _lib.ACT_NONE:
            gx = torch.empty_like(y)
            if y.numel():
                with torch.cuda.device(y.device):
                    _lib.check(_lib.load().hmp_bias_act_drop_bwd(g.data_ptr(), g.stride(0), y.data_ptr(), y.stride(0), y.size(0), y.size(1),
                                                                 ctx.act, ctx.p, gx.data_ptr(), gx.stride(0), _lib.stream_ptr()))
        else:
            gx = g
        gb = _colsum(gx) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gx, gb, None, None, None, None, None


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum: float, eps: float, training: bool):
        _dev(x, "x")
        x = _rows(x)
        n, F = x.shape
        y = torch.empty((n, F), dtype=torch.float32, device=x.device)
        save = torch.empty((2, F), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().hmp_batchnorm_fwd(x.data_ptr(), x.stride(0), n, F, gamma.data_ptr(), beta.data_ptr(),
                                                     running_mean.data_ptr(), running_var.data_ptr(), float(momentum), float(eps),
                                                     int(training), y.data_ptr(), y.stride(0), save.data_ptr(), _lib.stream_ptr()))
        ctx.training = training
        ctx.save_for_backward(x, gamma, save)
        ctx.mark_non_differentiable(running_mean, running_var)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, save = ctx.saved_tensors
        g = _rows(g)
        n, F = x.shape
        gx = torch.empty_like(x)
        gg, gb = torch.empty_like(gamma), torch.empty_like(gamma)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().hmp_batchnorm_bwd(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), n, F, gamma.data_ptr(),
                                                     save.data_ptr(), int(ctx.training), gx.data_ptr(), gx.stride(0), gg.data_ptr(),
                                                     gb.data_ptr(), _lib.stream_ptr()))
        return gx, gg, gb, None, None, None, None, None


class _SegmentMean(torch.autograd.Function):
    """``hmp_segment_mean_fwd`` / ``_bwd``: mean of the source rows over the incoming edges (LeafPool; zeros for rows
    without an edge)."""

    @staticmethod
    def forward(ctx, x, plan: GraphPlan):
        _dev(x, "x")
        x = _rows(x)
        if x.size(0) != plan.n_src:
            raise _lib.HydraMPError(f"segment_mean: {x.size(0)} rows for a plan over {plan.n_src} source nodes")
        ctx.plan = plan
        out = torch.empty((plan.n, x.size(1)), dtype=torch.float32, device=x.device)
        if out.numel():
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().hmp_segment_mean_fwd(x.data_ptr(), x.stride(0), x.size(1), plan.plan, out.data_ptr(),
                                                             out.stride(0), _lib.stream_ptr()))
        return out

    @staticmethod
    def backward(ctx, g):
        g = _rows(g)
        gx = torch.empty((ctx.plan.n_src, g.size(1)), dtype=torch.float32, device=g.device)
        if gx.numel() and ctx.plan.n == 0:
            gx.zero_()
        elif gx.numel():
            with torch.cuda.device(g.device):
                _lib.check(_lib.load().hmp_segment_mean_bwd(g.data_ptr(), g.stride(0), g.size(1), ctx.plan.plan, gx.data_ptr(),
                                                             gx.stride(0), _lib.stream_ptr()))
        return gx, None


def segment_mean(x: torch.Tensor, plan: GraphPlan) -> torch.Tensor:
    return _SegmentMean.apply(x, plan)


def project(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    return _Project.apply(x, weight)


def gcn_propagate(z: torch.Tensor, plan: GraphPlan) -> torch.Tensor:
    return _WSum.apply(z, plan, True, None)


def gin_propagate(z: torch.Tensor, plan: GraphPlan, eps: torch.Tensor) -> torch.Tensor:
    return _WSum.apply(z, plan, False, eps)


def bias_act_drop(x, bias=None, relu=False, p=0.0, seed=0, rng_step=0, rng_stream=0, elu=False) -> torch.Tensor:
    act = _lib.ACT_ELU if elu else (_lib.ACT_RELU if relu else _lib.ACT_NONE)
    return _BiasActDrop.apply(x, bias, act, p, seed, rng_step, rng_stream)


def batch_norm(x, gamma, beta, running_mean, running_var, momentum, eps, training) -> torch.Tensor:
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, momentum, eps, training)


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    """``x.argmax(dim=1)`` (first maximum) as int64 on the device (``hmp_argmax_rows``)"""
    _dev(x, "x")
    x = _rows(x)
    out = torch.empty(x.size(0), dtype=torch.int64, device=x.device)
    if x.size(0):
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().hmp_argmax_rows(x.data_ptr(), x.stride(0), x.size(0), x.size(1), out.data_ptr(), _lib.stream_ptr()))
    return out
