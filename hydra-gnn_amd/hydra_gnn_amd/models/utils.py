"""Model building utilities -- host-side mirror of the reference's ``src/hydra_gnn/models/utils.py``.

Same function names and argument meaning (``build_conv_layer`` :9-28, ``build_GAT_conv_layers``
:31-87, ``build_hetero_conv`` :90-100, ``build_GAT_hetero_conv`` :103-140, ``cross_entropy_loss``
:143-161), but the returned modules are *parameter containers* with torch_geometric's attribute
names (so ``state_dict()`` keys equal the reference's); the arithmetic runs in
``libhydra_mp.so`` through :class:`hydra_gnn_amd.engine.NativeNet`.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import _lib
from ..engine import ConvDesc
from .._lib import CONV_GAT, CONV_SAGE


class Linear(nn.Module):
    """Stand-in for ``torch_geometric.nn.dense.linear.Linear`` (keys ``weight`` / ``bias``)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, weight_initializer: Optional[str] = None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        if weight_initializer == "glorot":
            a = math.sqrt(6.0 / (in_channels + out_channels))
            nn.init.uniform_(self.weight, -a, a)
        else:
            b = 1.0 / math.sqrt(in_channels) if in_channels > 0 else 0.0
            nn.init.uniform_(self.weight, -b, b)
        if bias:
            b = 1.0 / math.sqrt(in_channels) if in_channels > 0 else 0.0
            nn.init.uniform_(self.bias, -b, b)


class SAGEConv(nn.Module):
    """Parameters of ``pyg_nn.SAGEConv(in, out, normalize=False, bias=True)``: ``lin_l`` (W_l, b_l) on the
    neighbour mean, ``lin_r`` (W_r, no bias) on the root."""

    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int, normalize: bool = False, bias: bool = True):
        super().__init__()
        if normalize or not bias:
            raise NotImplementedError("the reference builds SAGEConv(normalize=False, bias=True) only (models/utils.py:14)")
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = Linear(in_channels[0], out_channels, bias=True)
        self.lin_r = Linear(in_channels[1], out_channels, bias=False)

    def desc(self, edge_type) -> ConvDesc:
        return ConvDesc(CONV_SAGE, edge_type, self.out_channels,
                        {"w0": self.lin_l.weight, "b0": self.lin_l.bias, "w1": self.lin_r.weight})


class GATConv(nn.Module):
    """Parameters of ``pyg_nn.GATConv`` as the reference constructs it (models/utils.py:49-86)."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0.0,
                 add_self_loops=True, edge_dim=None, fill_value="mean", bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.heads, self.concat = in_channels, out_channels, heads, concat
        self.negative_slope, self.dropout, self.add_self_loops = negative_slope, dropout, add_self_loops
        self.edge_dim, self.fill_value = edge_dim, fill_value
        if abs(negative_slope - 0.2) > 1e-12:
            raise NotImplementedError("negative_slope is fixed at 0.2 (the reference never changes it)")
        if isinstance(in_channels, int):
            self.lin_src = Linear(in_channels, heads * out_channels, bias=False, weight_initializer="glorot")
            self.lin_dst = self.lin_src
            self.shared_lin = True
        else:
            self.lin_src = Linear(in_channels[0], heads * out_channels, bias=False, weight_initializer="glorot")
            self.lin_dst = Linear(in_channels[1], heads * out_channels, bias=False, weight_initializer="glorot")
            self.shared_lin = False
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        if edge_dim is not None:
            self.lin_edge = Linear(edge_dim, heads * out_channels, bias=False, weight_initializer="glorot")
            self.att_edge = nn.Parameter(torch.empty(1, heads, out_channels))
        else:
            self.lin_edge = None
            self.register_parameter("att_edge", None)
        if bias:
            self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels))
        else:
            raise NotImplementedError("GATConv(bias=False) is not used by the reference")
        a = math.sqrt(6.0 / (heads + out_channels))
        for p in (self.att_src, self.att_dst, self.att_edge):
            if p is not None:
                nn.init.uniform_(p, -a, a)

    def desc(self, edge_type) -> ConvDesc:
        if isinstance(self.fill_value, str):
            if self.fill_value != "mean":
                raise NotImplementedError("fill_value must be 'mean' or a tensor")
            fill_mean = 1
        else:
            fv = torch.as_tensor(self.fill_value)
            if bool((fv != 0).any()):
                raise NotImplementedError("tensor fill_value other than zeros is not used by the reference")
            fill_mean = 0
        return ConvDesc(
            CONV_GAT, edge_type, self.out_channels,
            {"w0": self.lin_src.weight, "w1": self.lin_dst.weight, "a0": self.att_src, "a1": self.att_dst,
             "w2": None if self.lin_edge is None else self.lin_edge.weight, "a2": self.att_edge, "b0": self.bias},
            heads=self.heads, concat=self.concat, self_loops=self.add_self_loops, edge_dim=self.edge_dim or 0,
            fill_mean=fill_mean, shared_lin=self.shared_lin, dropout=self.dropout,
        )


class HeteroConv(nn.Module):
    """Container with torch_geometric.nn.HeteroConv's parameter naming (``convs.<src>__<rel>__<dst>``)."""

    def __init__(self, convs: Dict[Tuple[str, str, str], nn.Module], aggr: str = "sum"):
        super().__init__()
        assert aggr in ("sum", "mean")
        self.edge_types = [tuple(k) for k in convs.keys()]
        self.convs = nn.ModuleDict({"__".join(k): v for k, v in convs.items()})
        self.aggr = aggr

    def conv(self, edge_type) -> nn.Module:
        return self.convs["__".join(edge_type)]

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # accept the PyG >= 2.4 ModuleDict spelling "<a___rel___b>" as well
        for et in self.edge_types:
            new = prefix + "convs.<" + "___".join(et) + ">."
            old = prefix + "convs." + "__".join(et) + "."
            for k in [k for k in state_dict if k.startswith(new)]:
                state_dict[old + k[len(new):]] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


class GCNConv(nn.Module):
    """Parameters of ``pyg_nn.GCNConv(in, out, add_self_loops=True)`` (reference models/utils.py:15-16): ``lin`` (glorot, no
    bias) and a zero-initialised ``bias``.  The arithmetic is ``ops.project`` -> ``ops.gcn_propagate`` -> ``ops.bias_act_drop``."""

    def __init__(self, in_channels: int, out_channels: int, add_self_loops: bool = True):
        super().__init__()
        if not add_self_loops:
            raise NotImplementedError("the reference builds GCNConv(add_self_loops=True) only (models/utils.py:16)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = Linear(in_channels, out_channels, bias=False, weight_initializer="glorot")
        self.bias = nn.Parameter(torch.zeros(out_channels))


class GINConv(nn.Module):
    """Parameters of ``pyg_nn.GINConv(nn, eps, train_eps)`` as the reference builds it (models/utils.py:17-26): ``nn`` is
    ``Sequential(Linear, ReLU, Linear)`` and ``eps`` a [1] parameter (``train_eps=True``)."""

    def __init__(self, mlp: nn.Module, eps: float = 0.0, train_eps: bool = False):
        super().__init__()
        if not (isinstance(mlp, nn.Sequential) and len(mlp) == 3 and isinstance(mlp[0], nn.Linear) and isinstance(mlp[1], nn.ReLU)
                and isinstance(mlp[2], nn.Linear)):
            raise NotImplementedError("GINConv: the native path runs the reference's Sequential(Linear, ReLU, Linear) only")
        self.nn = mlp
        self.initial_eps = eps
        if train_eps:
            self.eps = nn.Parameter(torch.tensor([float(eps)]))
        else:
            self.register_buffer("eps", torch.tensor([float(eps)]))


class BatchNorm(nn.Module):
    """``torch_geometric.nn.BatchNorm``: a wrapper around ``torch.nn.BatchNorm1d`` (state_dict keys ``module.*``); the
    arithmetic is ``ops.batch_norm``."""

    def __init__(self, in_channels: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.module = nn.BatchNorm1d(in_channels, eps=eps, momentum=momentum, affine=True, track_running_stats=True)


def build_conv_layer(conv_block, input_dim, output_dim, **kwargs):
    """reference models/utils.py:9-28"""
    if conv_block == "GraphSAGE":
        return SAGEConv(input_dim, output_dim, normalize=False, bias=True)
    if conv_block == "GCN":
        return GCNConv(input_dim, output_dim, add_self_loops=True)
    if conv_block == "GIN":
        return GINConv(nn.Sequential(nn.Linear(input_dim, output_dim), nn.ReLU(), nn.Linear(output_dim, output_dim)),
                       eps=0.0, train_eps=True)
    return NotImplemented  # the reference returns (not raises) NotImplemented for unknown blocks


def build_GAT_conv_layers(input_dim, hidden_dims, heads, concats, dropout=0.0, add_self_loop=True, edge_dim=None,
                          fill_value="mean"):
    """reference models/utils.py:31-87: layer i > 0 is built from an int width, so it shares lin_src/lin_dst."""
    assert len(hidden_dims) == len(heads)
    assert len(hidden_dims) == len(concats)
    convs = nn.ModuleList()
    fin = input_dim
    for i in range(len(hidden_dims)):
        convs.append(GATConv(fin, hidden_dims[i], heads=heads[i], concat=concats[i], dropout=dropout,
                             add_self_loops=add_self_loop, edge_dim=edge_dim, fill_value=fill_value))
        fin = hidden_dims[i] * heads[i] if concats[i] else hidden_dims[i]
    return convs


def build_hetero_conv(conv_block, edge_types, input_dim_dict, output_dim_dict, aggr="sum"):
    """reference models/utils.py:90-100."""
    conv_dict = dict()
    for source, edge_name, target in edge_types:
        conv_dict[source, edge_name, target] = build_conv_layer(
            conv_block, (input_dim_dict[source], input_dim_dict[target]), output_dim_dict[target])
    return HeteroConv(conv_dict, aggr=aggr)


def build_GAT_hetero_conv(edge_types, input_dim_dict, output_dim_dict, GAT_hidden_dims, GAT_heads, GAT_concats, dropout,
                          aggr="sum", edge_dim=None, fill_value="mean"):
    """reference models/utils.py:103-140."""
    chains = dict()
    for source, edge_name, target in edge_types:
        chains[source, edge_name, target] = build_GAT_conv_layers(
            (input_dim_dict[source], input_dim_dict[target]), GAT_hidden_dims + [output_dim_dict[target]], GAT_heads,
            GAT_concats, dropout, add_self_loop=(source == target), edge_dim=edge_dim, fill_value=fill_value)
    convs = nn.ModuleList()
    for i in range(len(GAT_heads)):
        convs.append(HeteroConv({et: chains[et][i] for et in chains}, aggr=aggr))
    return convs


class _MaskedCE(torch.autograd.Function):
    """``F.cross_entropy(pred[mask], label[mask])`` without the boolean-index sync: one native launch."""

    @staticmethod
    def forward(ctx, pred, label_eff):
        lib = _lib.require_device()
        pred = pred.contiguous()
        n, c = pred.shape
        grad = torch.empty_like(pred)
        out2 = torch.empty(2, dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            _lib.check(lib.hmp_masked_ce(pred.data_ptr(), pred.stride(0), n, c, label_eff.data_ptr(), -1, grad.data_ptr(),
                                         grad.stride(0), out2.data_ptr(), _lib.stream_ptr()))
        ctx.save_for_backward(grad, out2)
        return out2[0] / out2[1]

    @staticmethod
    def backward(ctx, g):
        grad, out2 = ctx.saved_tensors
        return grad * (g / out2[1]), None


def cross_entropy_loss(pred, label, mask=None):
    """reference models/utils.py:143-161 (tensor form = mean CE over ``mask``; list form = sum / count)."""
    if isinstance(pred, torch.Tensor):
        if not pred.is_cuda:
            raise _lib.HydraMPError("cross_entropy_loss: pred is on the CPU; hydra_gnn_amd has no CPU fallback")
        if pred.dtype != torch.float32:
            raise _lib.HydraMPError("cross_entropy_loss computes in fp32")
        label = label.to(torch.int64)
        if mask is not None:
            label = torch.where(mask, label, torch.full_like(label, -1))
        return _MaskedCE.apply(pred, label.contiguous())
    losses = [cross_entropy_loss(p, l, None if mask is None else m) for p, l, m in
              zip(pred, label, mask if mask is not None else [None] * len(pred))]
    counts = [torch.numel(l) if mask is None else m.sum() for l, m in zip(label, mask if mask is not None else label)]
    total = sum(c for c in counts)
    return sum(ls * c for ls, c in zip(losses, counts)) / total
