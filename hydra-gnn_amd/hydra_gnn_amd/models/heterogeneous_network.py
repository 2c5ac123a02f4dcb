"""``HeterogeneousNetwork`` -- drop-in for the reference's
``src/hydra_gnn/models/heterogeneous_network.py:13-136`` on the MI355X engine.

Constructor signature, ``forward(data)``, ``loss(pred, label, mask)`` and the ``state_dict`` keys
(``convs.{layer}.convs.{src}__{rel}__{dst}.lin_l.weight`` ...) are the reference's.  ``forward`` runs the
whole layer stack natively (``hmp_net_forward``), the backward of ``loss.backward()`` runs
``hmp_net_backward``; there is no PyG / ATen message-passing path and no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..data import EDGE_TYPES
from ..engine import LayerDesc, NativeNet
from .._lib import ACT_ELU, ACT_NONE, ACT_RELU
from .utils import build_GAT_hetero_conv, build_hetero_conv, cross_entropy_loss


class _NativeModule(nn.Module):
    """Shared plumbing: lazily built :class:`NativeNet`, dropout RNG bookkeeping."""

    def _init_native(self):
        self._native = None
        self._rng_step = 0
        # drawn from torch's default generator so torch.manual_seed() controls the dropout stream
        self._seed = int(torch.randint(0, 2 ** 62, (1,)).item())

    def native(self) -> NativeNet:
        if self._native is None:
            self._native = self._build_native()
        return self._native

    def _run(self, data) -> torch.Tensor:
        net = self.native()
        if self.training:
            self._rng_step += 1
        return net.forward(data, self.training, self._seed, self._rng_step)

    def _drop_stream(self, l: int, node_type: str) -> int:
        """RNG tensor id of the feature dropout after conv ``l`` on ``node_type`` (executor numbering: 8 * program layer + node
        type index; a ``pre_mp`` layer in front shifts the program layers by one)"""
        net = self.native()
        return (l + len(net.layers) - self.num_layers) * 8 + net.node_types.index(node_type)

    def _tail_act_drop(self, x, stream: int):
        """activation + dropout on a final state of the two-headed task (reference heterogeneous_network.py:124-134): the native
        program ends at the last conv, this is ``hmp_bias_act_drop_*`` on its output; keep-mask = tensor ``stream`` of this call's
        RNG step (layer L-1 of the executor's numbering, which the executor itself never draws: its last layer has no dropout)"""
        from .. import ops

        gat = self.conv_block[:3] == "GAT"
        return ops.bias_act_drop(x, None, relu=not gat, elu=gat, p=self.dropout if self.training else 0.0, seed=self._seed,
                                 rng_step=self._rng_step, rng_stream=stream)

    def train_step(self, lr, weight_decay=0.0, **kw):
        """Fused native training step (fwd + masked CE + bwd [+ all-reduce] + Adam), see engine.TrainStep."""
        from ..engine import TrainStep

        if getattr(self, "classification_task", "room") != "room":
            raise NotImplementedError("the fused training step computes ONE masked cross entropy (room labels); the two-headed "
                                      "task trains through loss.backward() like the reference's SemiSupervisedTrainingJob")
        return TrainStep(self.native(), lr=lr, weight_decay=weight_decay, view=getattr(self, "_view", None), **kw)

    def predict(self, data):
        """Room labels of ``data`` on the host: ``self(data).argmax(dim=1).cpu()`` as ``GnnModel.infer`` computes it
        (bin/room_classification_server:285-286), through the native inference path (:meth:`NativeNet.predict`)."""
        net = self.native()
        view = getattr(self, "_view", None)
        labels = net.predict(view(data) if view is not None else data, net.layers[-1].out_dims[net.readout])
        mask = getattr(data, "room_mask", None) if view is not None else None
        return labels if mask is None else labels[mask.cpu()]

    def loss(self, pred, label, mask=None):
        return cross_entropy_loss(pred, label, mask)


def _hetero_layers(module, node_types):
    """LayerDesc list from ``module.convs`` (a ModuleList of HeteroConv containers)."""
    gat = module.conv_block[:3] == "GAT"
    layers = []
    L = module.num_layers
    for l, hc in enumerate(module.convs):
        convs = [hc.conv(et).desc(et) for et in hc.edge_types]
        out_dims = {}
        for et, cd in zip(hc.edge_types, convs):
            c = hc.conv(et)
            width = cd.f_out * (c.heads if gat and c.concat else 1)
            assert out_dims.setdefault(et[2], width) == width, "convs reaching one node type must agree on the width"
        last = l == L - 1
        act = ACT_NONE if last else (ACT_ELU if gat else ACT_RELU)
        layers.append(LayerDesc(convs, out_dims, act, 0.0 if last else module.dropout, group_mean=(hc.aggr == "mean")))
    return layers


class HeterogeneousNetwork(_NativeModule):
    def __init__(
        self,
        input_dim_dict,
        output_dim=None,
        output_dim_dict=None,
        conv_block="GraphSAGE",
        hidden_dim=None,
        num_layers=None,
        GAT_hidden_dims=None,
        GAT_heads=None,
        GAT_concats=None,
        dropout=0.25,
        **kwargs
    ):
        """Arguments as in the reference (``heterogeneous_network.py:27-39``); ``**kwargs`` swallows
        ``ignored_label`` etc. exactly like the reference does."""
        super().__init__()
        assert conv_block in ["GraphSAGE", "GAT", "GAT_edge"]
        self.conv_block = conv_block
        if output_dim is not None:
            assert output_dim_dict is None
            self.classification_task = "room"
            output_dim_dict = {"rooms": output_dim, "objects": output_dim}  # final objects states are ignored
        else:
            assert output_dim_dict is not None
            self.classification_task = "all"  # two outputs: the executor's second readout (hmp_net_aux_output / hmp_net_backward2)
        self.num_layers = num_layers if conv_block[:3] != "GAT" else len(GAT_heads)
        self.dropout = dropout
        self.input_dim_dict = dict(input_dim_dict)

        hidden_dim_dict = {"rooms": hidden_dim, "objects": hidden_dim}
        if conv_block == "GAT":
            self.convs = build_GAT_hetero_conv(EDGE_TYPES, input_dim_dict, output_dim_dict, GAT_hidden_dims, GAT_heads,
                                               GAT_concats, dropout)
        elif conv_block == "GAT_edge":
            self.convs = build_GAT_hetero_conv(EDGE_TYPES, input_dim_dict, output_dim_dict, GAT_hidden_dims, GAT_heads,
                                               GAT_concats, dropout, edge_dim=3,
                                               fill_value=torch.zeros(3, dtype=torch.float64))
        else:
            dims = [input_dim_dict] + [hidden_dim_dict] * (self.num_layers - 1) + [output_dim_dict]
            self.convs = nn.ModuleList(
                build_hetero_conv(conv_block, EDGE_TYPES, dims[l], dims[l + 1]) for l in range(self.num_layers))
        self._init_native()

    def _build_native(self) -> NativeNet:
        node_types = ["objects", "rooms"]
        return NativeNet(node_types, self.input_dim_dict, EDGE_TYPES, _hetero_layers(self, node_types), readout="rooms",
                         aux_readout="objects" if self.classification_task == "all" else None)

    def forward(self, data):
        out = self._run(data)
        dims = self.native().layers[-1].out_dims
        if self.classification_task == "room":
            return out[:, : dims["rooms"]]
        rooms, objects = out
        last = self.num_layers - 1
        return (self._tail_act_drop(rooms[:, : dims["rooms"]], self._drop_stream(last, "rooms")),
                self._tail_act_drop(objects[:, : dims["objects"]], self._drop_stream(last, "objects")))
