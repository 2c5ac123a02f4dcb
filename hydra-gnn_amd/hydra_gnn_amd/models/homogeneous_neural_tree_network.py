"""``HomogeneousNeuralTreeNetwork`` -- drop-in for the reference's
``src/hydra_gnn/models/homogeneous_neural_tree_network.py:7-109`` on the MI355X engine.

One node type, three edge types of ONE native program: ``pre_mp`` (a 1-head GAT over ``init_edge_index`` without self
loops; the reference applies it to EVERY node, so nodes without an incoming initialisation edge end up with the bias
alone, :83-84), the message passing layers over ``edge_index``, and ``LeafPool`` (mean over ``pool_edge_index``, the
executor's pool stage).  The readout ``x[room_mask]`` stays in torch.

GCN / GIN (``htree_GCN.yaml`` / ``htree_GIN.yaml``): ``pre_mp`` runs as a one-layer native program, the convolutions and
``LeafPool`` op by op (:mod:`hydra_gnn_amd.ops`); as in the reference, this class's loop never applies ``batch_norms``.
"""
from __future__ import annotations

import torch.nn as nn

from .. import ops
from ..engine import LayerDesc, NativeNet
from .._lib import ACT_NONE
from .heterogeneous_neural_tree_network import LeafPool
from .homogeneous_network import _EDGE, _NODE, HomogeneousNetwork
from .utils import GATConv

_INIT = (_NODE, "init", _NODE)
_POOL = (_NODE, "pool", _NODE)


class _HtreeView:
    """homogeneous H-tree ``Data`` through the hetero accessors (edge types: message passing, init, pool)"""

    def __init__(self, data, with_init):
        self._d, self._init = data, with_init

    @property
    def x_dict(self):
        return {_NODE: self._d.x}

    @property
    def edge_index_dict(self):
        d = {_EDGE: self._d.edge_index, _POOL: self._d.pool_edge_index}
        if self._init:
            d[_INIT] = self._d.init_edge_index
        return d

    @property
    def edge_attr_dict(self):
        ea = getattr(self._d, "edge_attr", None)
        return {} if ea is None else {_EDGE: ea}

    def __getitem__(self, key):
        class _N:
            num_nodes = int(self._d.x.size(0))

        return _N


class _InitView:
    """the ``pre_mp`` program's view: x and the initialisation edges only"""

    def __init__(self, data):
        self._d = data

    @property
    def x_dict(self):
        return {_NODE: self._d.x}

    @property
    def edge_index_dict(self):
        return {_INIT: self._d.init_edge_index}

    @property
    def edge_attr_dict(self):
        return {}


class HomogeneousNeuralTreeNetwork(HomogeneousNetwork):
    def __init__(
        self,
        input_dim,
        output_dim=None,
        output_dim_dict=None,
        conv_block="GCN",
        disable_initialization=False,
        hidden_dim=None,
        num_layers=None,
        GAT_hidden_dims=None,
        GAT_heads=None,
        GAT_concats=None,
        dropout=0.25,
        **kwargs
    ):
        super().__init__(input_dim, output_dim, output_dim_dict, conv_block, hidden_dim, num_layers, GAT_hidden_dims, GAT_heads,
                         GAT_concats, dropout, **kwargs)
        if disable_initialization:
            self.pre_mp = None
            print("diable initialization")
        else:
            if input_dim > 256:
                raise NotImplementedError(
                    f"pre_mp over {input_dim}-wide features: the GAT kernels hold one 4-channel slice per lane (<= 256 channels per "
                    "head).  Every shipped H-tree config sets disable_initialization: True; the 6-d Stanford / --remove_word2vec "
                    "features are supported.")
            self.pre_mp = GATConv(input_dim, input_dim, heads=1, concat=False, dropout=0.0, add_self_loops=False)
        self.post_mp_pool = LeafPool(aggr="mean")
        self._native = None  # the parent built no program yet (lazy), but make the rebuild explicit

    def _build_native(self) -> NativeNet:
        if self.op_path:  # GCN / GIN: the program holds pre_mp alone (the input has no gradient, so it can be a prefix)
            init = LayerDesc([self.pre_mp.desc(_INIT)], {_NODE: self.input_dim}, ACT_NONE, 0.0)
            return NativeNet([_NODE], {_NODE: self.input_dim}, [_INIT], [init], readout=_NODE)
        base = super()._build_native()
        layers = list(base.layers)
        edge_types = [_EDGE, _POOL]
        if self.pre_mp is not None:
            init = LayerDesc([self.pre_mp.desc(_INIT)], {_NODE: self.input_dim}, ACT_NONE, 0.0)
            layers = [init] + layers
            edge_types.append(_INIT)
        return NativeNet([_NODE], {_NODE: self.input_dim}, edge_types, layers, readout=_NODE, pool_edge_type=_POOL)

    def _drop_stream(self, l: int) -> int:
        # with pre_mp inside the native program (SAGE / GAT) the convs are its layers 1..L
        return 8 * (l + (1 if (self.pre_mp is not None and not self.op_path) else 0))

    def _view(self, data):
        return _HtreeView(data, self.pre_mp is not None)

    def forward(self, data):
        if self.op_path:
            x = data.x
            if self.pre_mp is not None:
                x = self.native().forward(_InitView(data), self.training, self._seed, 0)[:, : self.input_dim]
            n = data.x.size(0)
            x = self._op_layers(x, ops.GraphPlan(data.edge_index, n), batch_norm=False)
            x = ops.segment_mean(x, ops.GraphPlan(data.pool_edge_index, n))
            return self._op_heads(x, data.room_mask, getattr(data, "object_mask", None))
        out = self._run(_HtreeView(data, self.pre_mp is not None))
        out = out[:, : self.native().layers[-1].out_dims[_NODE]]
        return self._op_heads(out, data.room_mask, getattr(data, "object_mask", None))  # reference :96-109
