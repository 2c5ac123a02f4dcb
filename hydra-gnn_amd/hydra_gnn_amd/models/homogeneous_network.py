"""``HomogeneousNetwork`` -- drop-in for the reference's ``src/hydra_gnn/models/homogeneous_network.py:11-147`` on the
MI355X engine.

GraphSAGE / GAT / GAT_edge: a homogeneous graph is the one-node-type, one-edge-type case of the native program
(:class:`hydra_gnn_amd.engine.NativeNet`, one launch sequence per step).  GCN / GIN (+ BatchNorm) -- the Stanford3DSG
``baseline_GCN`` / ``baseline_GIN`` configurations, SURVEY.md 8(f) row 2 -- are composed op by op from the native operators
of :mod:`hydra_gnn_amd.ops` (their graphs have 2..27 nodes; the per-layer launches do not matter there).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..engine import LayerDesc, NativeNet
from .._lib import ACT_ELU, ACT_NONE, ACT_RELU, HydraMPError
from .heterogeneous_network import _NativeModule
from .utils import BatchNorm, build_conv_layer, build_GAT_conv_layers

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
        if conv_block not in ("GraphSAGE", "GAT", "GAT_edge", "GCN", "GIN"):
            raise NotImplementedError(f"conv_block {conv_block}")
        self.conv_block = conv_block
        self.op_path = conv_block in ("GCN", "GIN")
        if output_dim is None:  # two-headed task of the semi-supervised Stanford job (reference :57-63,99-120)
            assert output_dim_dict is not None
            self.classification_task = "all"
            output_dim = hidden_dim
        else:
            assert output_dim_dict is None
            self.classification_task = "room"
        gat = conv_block[:3] == "GAT"
        self.num_layers = num_layers if not gat else len(GAT_heads)
        self.dropout = dropout
        self.input_dim = input_dim
        all_task = self.classification_task == "all"
        gat_dims = (list(GAT_hidden_dims) if all_task else list(GAT_hidden_dims) + [output_dim]) if gat else None
        if conv_block == "GAT":
            self.convs = build_GAT_conv_layers(input_dim, gat_dims, GAT_heads, GAT_concats, dropout=dropout)
        elif conv_block == "GAT_edge":
            self.convs = build_GAT_conv_layers(input_dim, gat_dims, GAT_heads, GAT_concats,
                                               dropout=dropout, edge_dim=3, add_self_loop=True,
                                               fill_value=torch.zeros(3, dtype=torch.float64))
        else:
            dims = [input_dim] + [hidden_dim] * (self.num_layers - 1) + [output_dim]
            self.convs = nn.ModuleList(build_conv_layer(conv_block, dims[l], dims[l + 1]) for l in range(self.num_layers))
        if conv_block == "GIN":  # reference :93-97 (one per layer, the last one is never used)
            self.batch_norms = nn.ModuleList(BatchNorm(hidden_dim) for _ in range(self.num_layers))
        if self.classification_task == "all":  # reference :99-120
            n_room = output_dim_dict["rooms"] if "rooms" in output_dim_dict else output_dim_dict["room"]
            n_obj = output_dim_dict["objects"] if "objects" in output_dim_dict else output_dim_dict["object"]
            final_hidden = (gat_dims[-1] * GAT_heads[-1] if GAT_concats[-1] else gat_dims[-1]) if gat else hidden_dim
            self.post_mp_room = nn.Linear(final_hidden, n_room)
            self.post_mp_object = nn.Linear(final_hidden, n_obj)
        self._init_native()

    def _build_native(self) -> NativeNet:
        if self.op_path:
            raise HydraMPError(f"{self.conv_block} runs op by op (hydra_gnn_amd.ops); there is no fused program / train_step "
                               "for it: use loss.backward() and an optimiser as the reference's training loop does")
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

    # ---- GCN / GIN: op-by-op native path ---------------------------------------------------------------------------------
    def _drop_stream(self, l: int) -> int:
        """RNG tensor id of the feature dropout after conv ``l`` (the native program numbers its layers from 0, node type 0)"""
        return 8 * l

    def _act_drop(self, x, l, bias=None):
        """activation (relu; elu for GAT) + dropout after layer ``l`` (reference :135-136,141-142); the keep-mask is tensor
        ``_drop_stream(l)`` of this call's RNG step"""
        p = self.dropout if self.training else 0.0
        gat = self.conv_block[:3] == "GAT"
        return ops.bias_act_drop(x, bias, relu=not gat, elu=gat, p=p, seed=self._seed, rng_step=self._rng_step,
                                 rng_stream=self._drop_stream(l))

    def _op_layers(self, x, plan, batch_norm=True):
        """reference :125-136 for conv_block GCN / GIN; ``batch_norm`` False = the H-tree variant's loop, which never
        applies ``batch_norms`` (homogeneous_neural_tree_network.py:86-94)."""
        if self.training:
            self._rng_step += 1
        L = self.num_layers
        for l, conv in enumerate(self.convs):
            last = l == L - 1
            if self.conv_block == "GCN":
                a = ops.gcn_propagate(ops.project(x, conv.lin.weight), plan)
                x = ops.bias_act_drop(a, conv.bias) if last else self._act_drop(a, l, conv.bias)
            else:
                a = ops.gin_propagate(ops.project(x, conv.nn[0].weight), plan, conv.eps)
                h = ops.bias_act_drop(a, conv.nn[0].bias, relu=True)
                y = ops.bias_act_drop(ops.project(h, conv.nn[2].weight), conv.nn[2].bias)
                if last:
                    x = y
                else:
                    if batch_norm:
                        bn = self.batch_norms[l].module
                        y = ops.batch_norm(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, self.training)
                        if self.training:
                            bn.num_batches_tracked += 1
                    x = self._act_drop(y, l)
        return x

    def _op_heads(self, x, room_mask, object_mask):
        """reference :138-146"""
        if self.classification_task == "room":
            return x[room_mask, :]
        x = self._act_drop(x, self.num_layers - 1)
        head = lambda lin, rows: ops.bias_act_drop(ops.project(rows, lin.weight), lin.bias)
        return head(self.post_mp_room, x[room_mask, :]), head(self.post_mp_object, x[object_mask, :])

    def predict(self, data):
        if self.classification_task != "room":
            raise NotImplementedError("predict() returns room labels (the server's task)")
        if not self.op_path:
            return super().predict(data)
        with torch.no_grad():
            return ops.argmax_rows(self(data)).cpu()

    def forward(self, data):
        if self.op_path:
            plan = ops.GraphPlan(data.edge_index, data.x.size(0))
            x = self._op_layers(data.x, plan)
            return self._op_heads(x, data.room_mask, ~data.room_mask)
        out = self._run(_HomoView(data))
        out = out[:, : self.native().layers[-1].out_dims[_NODE]]
        return self._op_heads(out, data.room_mask, ~data.room_mask)
