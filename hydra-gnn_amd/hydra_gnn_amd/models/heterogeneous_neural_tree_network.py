"""``HeterogeneousNeuralTreeNetwork`` -- drop-in for the reference's
``src/hydra_gnn/models/heterogeneous_neural_tree_network.py:34-205`` on the MI355X engine.

Message passing over the 10 ``HTREE_EDGE_TYPES`` of an augmented H-tree, then ``LeafPool`` (mean of the
``room`` leaves of every ``room_virtual`` node, reference :18-31,182-185) -- all inside one native program
(the pool is the executor's ``pool_edge_type`` stage).  ``pre_mp`` (GAT initialisation of the clique
nodes, reference :92-103) is built when ``disable_initialization=False``; every shipped H-tree config
disables it (``config/mp3d/htree_gt60.yaml:10``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..data import HTREE_EDGE_TYPES, HTREE_INIT_EDGE_TYPES, HTREE_NODE_TYPES
from ..engine import LayerDesc, NativeNet
from .._lib import ACT_NONE
from .heterogeneous_network import _NativeModule, _hetero_layers
from .utils import GATConv, HeteroConv, build_GAT_hetero_conv, build_hetero_conv

POOL_EDGE_TYPE = ("room", "r_to_rv", "room_virtual")


class LeafPool(nn.Module):
    """Parameter-free marker module (keeps ``post_mp`` in the module tree like the reference)."""

    def __init__(self, aggr="mean"):
        super().__init__()
        assert aggr == "mean"


class HeterogeneousNeuralTreeNetwork(_NativeModule):
    def __init__(
        self,
        input_dim_dict,
        output_dim=None,
        output_dim_dict=None,
        conv_block="GraphSAGE",
        disable_initialization=False,
        hidden_dim=None,
        num_layers=None,
        GAT_hidden_dims=None,
        GAT_heads=None,
        GAT_concats=None,
        dropout=0.25,
        **kwargs
    ):
        super().__init__()
        assert conv_block in ["GraphSAGE", "GAT", "GAT_edge"]
        self.conv_block = conv_block
        if output_dim is not None:
            assert output_dim_dict is None
            self.classification_task = "room"
            output_dim_dict = {node_type: output_dim for node_type in HTREE_NODE_TYPES}
        else:
            assert output_dim_dict is not None
            self.classification_task = "all"  # two outputs (reference :186-205): second readout + LeafPool per head on the operators
        self.num_layers = num_layers if conv_block[:3] != "GAT" else len(GAT_heads)
        self.dropout = dropout

        assert input_dim_dict["object"] == input_dim_dict["object_virtual"]
        assert input_dim_dict["room"] == input_dim_dict["room_virtual"]
        assert input_dim_dict["object-room"] == input_dim_dict["room-room"]
        self.input_dim_dict = dict(input_dim_dict)

        if disable_initialization:
            self.pre_mp = None
            print("diable initialization")
        else:
            self.pre_mp = HeteroConv(
                {
                    (s, r, t): GATConv((input_dim_dict[s], input_dim_dict[t]), input_dim_dict[t], heads=1, concat=False,
                                       dropout=0.0, add_self_loops=False)
                    for s, r, t in HTREE_INIT_EDGE_TYPES
                },
                aggr="mean",
            )

        mp_in = {t: input_dim_dict[t] for t in HTREE_NODE_TYPES}
        hidden = {t: hidden_dim for t in HTREE_NODE_TYPES}
        if conv_block == "GAT":
            self.convs = build_GAT_hetero_conv(HTREE_EDGE_TYPES, mp_in, output_dim_dict, GAT_hidden_dims, GAT_heads,
                                               GAT_concats, dropout)
        elif conv_block == "GAT_edge":
            self.convs = build_GAT_hetero_conv(HTREE_EDGE_TYPES, mp_in, output_dim_dict, GAT_hidden_dims, GAT_heads,
                                               GAT_concats, dropout, edge_dim=3,
                                               fill_value=torch.zeros(3, dtype=torch.float64))
        else:
            dims = [mp_in] + [hidden] * (self.num_layers - 1) + [output_dim_dict]
            self.convs = nn.ModuleList(
                build_hetero_conv(conv_block, HTREE_EDGE_TYPES, dims[l], dims[l + 1]) for l in range(self.num_layers))
        self.post_mp = LeafPool(aggr="mean")
        self._init_native()

    def _build_native(self) -> NativeNet:
        layers = _hetero_layers(self, HTREE_NODE_TYPES)
        two = self.classification_task == "all"
        pool = dict(readout="room", aux_readout="object") if two else dict(readout="room", pool_edge_type=POOL_EDGE_TYPE)
        pool_et = [] if two else [POOL_EDGE_TYPE]
        if self.pre_mp is None:
            if two:
                return NativeNet(list(HTREE_NODE_TYPES), {t: self.input_dim_dict[t] for t in HTREE_NODE_TYPES},
                                 list(HTREE_EDGE_TYPES), layers, **pool)
            node_types = list(HTREE_NODE_TYPES) + ["room_virtual"]
            in_dims = {t: self.input_dim_dict[t] for t in HTREE_NODE_TYPES}
            return NativeNet(node_types, in_dims, list(HTREE_EDGE_TYPES) + [POOL_EDGE_TYPE], layers,
                             readout="room", pool_edge_type=POOL_EDGE_TYPE, count_types=["room_virtual"])
        # pre_mp (reference :92-103,158-159) = one more GAT layer in front: the virtual nodes (copies of the scene-graph
        # nodes) write the clique features, the leaves keep theirs (`x_dict.update(...)` -> passthrough)
        convs = [self.pre_mp.conv(et).desc(et) for et in self.pre_mp.edge_types]
        out_dims = {et[2]: self.input_dim_dict[et[2]] for et in self.pre_mp.edge_types}
        leaves = [t for t in HTREE_NODE_TYPES if t not in out_dims]
        out_dims.update({t: self.input_dim_dict[t] for t in leaves})
        init = LayerDesc(convs, out_dims, ACT_NONE, 0.0, group_mean=(self.pre_mp.aggr == "mean"), passthrough=leaves)
        node_types = list(HTREE_NODE_TYPES) + ["room_virtual", "object_virtual"]
        in_dims = {t: self.input_dim_dict[t] for t in node_types}
        return NativeNet(node_types, in_dims, list(HTREE_EDGE_TYPES) + pool_et + list(HTREE_INIT_EDGE_TYPES),
                         [init] + layers, **pool)

    def forward(self, data):
        out = self._run(data)
        dims = self.native().layers[-1].out_dims
        if self.classification_task == "room":
            return out[:, : dims["room"]]
        # reference :186-205: activation + dropout on the final states, then LeafPool per head, first N_virtual rows
        from .. import ops

        heads = []
        for x, leaf, rel, virt in ((out[0], "room", "r_to_rv", "room_virtual"), (out[1], "object", "o_to_ov", "object_virtual")):
            h = self._tail_act_drop(x[:, : dims[leaf]], self._drop_stream(self.num_layers - 1, leaf))
            n_virtual = int(data[virt].num_nodes)
            plan = ops.GraphPlan(data[leaf, rel, virt].edge_index, n_virtual, num_src=h.size(0))
            heads.append(ops.segment_mean(h, plan))
        return tuple(heads)
