"""``HomogeneousNetwork`` (GraphSAGE / GAT / GAT_edge branches) -- drop-in for the reference's
``src/hydra_gnn/models/homogeneous_network.py:11-147`` on the MI355X engine.

A homogeneous graph is the one-node-type, one-edge-type case of the native program.  GCN and GIN are
outside the hot path and raise ``NotImplementedError``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..engine import LayerDesc, NativeNet
from .._lib import ACT_ELU, ACT_NONE, ACT_RELU
from .heterogeneous_network import _NativeModule
from .utils import build_conv_layer, build_GAT_conv_layers

_NODE = "node"
_EDGE = (_NODE, "to", _NODE)


class _HomoView:
    """Presents a homogeneous ``Data`` (x, edge_index, edge_attr) through the hetero accessors."""

    def __init__(self, data):
        self._d = data

    @property
    def x_dict(self):
        return {_NODE: self._d.x}

    @property
    def edge_index_dict(self):
        return {_EDGE: self._d.edge_index}

    @property
    def edge_attr_dict(self):
        ea = getattr(self._d, "edge_attr", None)
        return {} if ea is None else {_EDGE: ea}


class HomogeneousNetwork(_NativeModule):
    def __init__(
        self,
        input_dim,
        output_dim=None,
        output_dim_dict=None,
        conv_block="GCN",
        hidden_dim=None,
        num_layers=None,
        GAT_hidden_dims=None,
        GAT_heads=None,
        GAT_concats=None,
        dropout=0.25,
        **kwargs
    ):
        super().__init__()
        if conv_block not in ("GraphSAGE", "GAT", "GAT_edge"):
            raise NotImplementedError(f"conv_block {conv_block}: only GraphSAGE / GAT / GAT_edge are on the MI355X hot path")
        self.conv_block = conv_block
        if output_dim is None:
            raise NotImplementedError("classification_task='all' is outside the MI355X hot path (SURVEY.md section 2, row 8)")
        assert output_dim_dict is None
        self.classification_task = "room"
        gat = conv_block[:3] == "GAT"
        self.num_layers = num_layers if not gat else len(GAT_heads)
        self.dropout = dropout
        self.input_dim = input_dim
        if conv_block == "GAT":
            self.convs = build_GAT_conv_layers(input_dim, GAT_hidden_dims + [output_dim], GAT_heads, GAT_concats, dropout=dropout)
        elif conv_block == "GAT_edge":
            self.convs = build_GAT_conv_layers(input_dim, GAT_hidden_dims + [output_dim], GAT_heads, GAT_concats,
                                               dropout=dropout, edge_dim=3, add_self_loop=True,
                                               fill_value=torch.zeros(3, dtype=torch.float64))
        else:
            dims = [input_dim] + [hidden_dim] * (self.num_layers - 1) + [output_dim]
            self.convs = nn.ModuleList(build_conv_layer(conv_block, dims[l], dims[l + 1]) for l in range(self.num_layers))
        self._init_native()

    def _build_native(self) -> NativeNet:
        gat = self.conv_block[:3] == "GAT"
        layers = []
        for l, conv in enumerate(self.convs):
            cd = conv.desc(_EDGE)
            width = cd.f_out * (conv.heads if gat and conv.concat else 1)
            last = l == self.num_layers - 1
            act = ACT_NONE if last else (ACT_ELU if gat else ACT_RELU)
            layers.append(LayerDesc([cd], {_NODE: width}, act, 0.0 if last else self.dropout))
        return NativeNet([_NODE], {_NODE: self.input_dim}, [_EDGE], layers, readout=_NODE)

    def _view(self, data):
        return _HomoView(data)

    def forward(self, data):
        out = self._run(_HomoView(data))
        out = out[:, : self.native().layers[-1].out_dims[_NODE]]
        return out[data.room_mask, :]
