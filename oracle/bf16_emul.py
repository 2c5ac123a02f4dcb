"""ORACLE (test infrastructure, not product code) -- BASELINE config 5's precision contract ("hidden=256 bf16": bf16 storage /
fp32 accumulate) restated on the CPU: the 3-layer HeteroConv(SAGE) room classifier of
``src/hydra_gnn/models/heterogeneous_network.py:80-122`` (SAGEConv + HeteroConv(sum) semantics as in :mod:`oracle.pyg_ref`)
evaluated in float64 (or float32) with a round-to-nearest-even to bfloat16 at EXACTLY the points where the engine's bf16 mode
rounds:

forward, layer l
  * GEMM operands: the layer input H[l] and the weights (the root weight is ``sum_e W_r,e`` summed in fp32 FIRST, then rounded);
    products accumulate unrounded;
  * the projected rows Z[l] of a 256-wide layer are STORED as bf16 (neighbour blocks and root block); the last (26-wide) layer
    keeps them unrounded;
  * neighbour mean, + root + bias, ReLU, dropout (scale 1/(1-p)) unrounded; the hidden activations H[l+1] are STORED as bf16;
backward, layer l
  * the gradient G[l+1] of a hidden layer's pre-activation (after the ReLU / dropout mask and scale) is STORED as bf16; it is
    the root block of dZ[l] as it stands; the neighbour blocks of dZ[l] (transposed mean of G) are STORED as bf16;
  * last layer: dZ is stored unrounded but ROUNDED as the operand of both backward GEMMs (input gradient, weight / bias gradient);
  * weight gradient = dZ^T [H | 1] with both operands as rounded above (the bias gradient is the ones column: column sums of the
    rounded root block).

aggregate-first convs (``agg_first``: the (layer, edge type) pairs the engine evaluates in [PyG] SAGEConv's own order, mean of the
source rows first -- objects -> rooms at >= 32 768 objects, ``hmp_conv_spec.agg_first``)
  * the mean M of the source rows AS STORED (the fp32 input features in layer 0, bf16 activations later) is kept in fp32 and ROUNDED as the operand of the destination-sized GEMM M * W_l^T;
    that product is the conv's block of Z[l][dst] (stored as bf16 in a hidden layer); backward: its block of dZ is the destination's
    output gradient (as stored), dM = dZ_block * W_l stays fp32 and reaches the source rows BEFORE their activation mask;

Because the rounding points depend on the engine's algebra (project first, then aggregate: SURVEY App. C.3), this restatement
follows that algebra; with every rounding removed (``rounding=False``) it equals ``oracle.models.HeterogeneousNetwork`` up to
float64 round-off, which ``tests/test_oracle_kat.py`` checks -- that is what ties it to the PyG restatement.

Parity status: as ``pyg_ref.py`` ("parity unpinned": no reference-held vector pins a model output).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from .models import EDGE_TYPES


def rnd_bf16(x: torch.Tensor) -> torch.Tensor:
    """round to nearest even bfloat16, returned in x's dtype"""
    return x.to(torch.float32).to(torch.bfloat16).to(x.dtype)


class _RoundFwd(torch.autograd.Function):
    """value rounded, gradient passed through (a stored / operand rounding of the forward pass)"""

    @staticmethod
    def forward(ctx, x):
        return rnd_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """value passed through, gradient rounded (a stored / operand rounding of the backward pass)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return rnd_bf16(g)


def _scatter_mean(msg, index, n):
    out = msg.new_zeros((n, msg.size(1))).index_add_(0, index, msg)
    cnt = torch.bincount(index, minlength=n).clamp(min=1).to(msg.dtype)
    return out / cnt[:, None]


def sage_hetero_bf16(ora, batch, dtype=torch.float64, rounding: bool = True,
                     dropout_fn: Optional[Callable] = None, training: bool = False, agg_first=()):
    """(logits [N_rooms, C], loss, {parameter name: gradient}) of ``ora`` (an ``oracle.models.HeterogeneousNetwork`` with
    GraphSAGE convs, classification_task 'room') on ``batch`` under the bf16 storage contract above.  ``dropout_fn(x, p,
    training, tag)`` as in ``oracle.models`` (replays the engine's keep-masks)."""
    fwd = _RoundFwd.apply if rounding else (lambda t: t)
    bwd = _RoundBwd.apply if rounding else (lambda t: t)
    L = ora.num_layers
    params = {n: p for n, p in ora.named_parameters()}
    leaves: Dict[str, torch.Tensor] = {}

    def leaf(name):  # fp32 parameter as a differentiable leaf
        if name not in leaves:
            leaves[name] = params[name].detach().clone().requires_grad_(True)
        return leaves[name]

    x = {t: batch[t].x.detach().to(dtype) for t in ("objects", "rooms")}
    ei = {et: batch[et].edge_index for et in EDGE_TYPES}
    for l in range(L):
        last = l == L - 1
        dsts = ["rooms"] if last else ["objects", "rooms"]  # last layer: convs into `objects` never reach the readout
        h_in = {s: fwd(x[s]) for s in x}
        nxt = {}
        for t in dsts:
            convs_t = [et for et in EDGE_TYPES if et[2] == t]
            key = lambda et: f"convs.{l}.convs.{'__'.join(et)}"
            w_root32 = None
            bias32 = None
            for et in convs_t:  # summed in fp32 (the pack kernel), THEN rounded as a GEMM operand
                wr, b = leaf(key(et) + ".lin_r.weight"), leaf(key(et) + ".lin_l.bias")
                w_root32 = wr if w_root32 is None else w_root32 + wr
                bias32 = b if bias32 is None else bias32 + b
            zroot = h_in[t] @ fwd(w_root32.to(dtype)).t()
            if not last:
                zroot = fwd(zroot)  # root block of Z stored as bf16
            agg = 0
            for et in convs_t:
                w_l = fwd(leaf(key(et) + ".lin_l.weight").to(dtype))
                if (l, tuple(et)) in agg_first:  # mean of the STORED source rows first (fp32 input features in layer 0, bf16
                    # activations after it: x[] as it stands), then the destination-sized projection
                    m = _scatter_mean(x[et[0]].index_select(0, ei[et][0]), ei[et][1], x[t].size(0))
                    z = fwd(m) @ w_l.t()
                    z = bwd(fwd(z)) if not last else bwd(z)
                    agg = agg + z
                    continue
                z = h_in[et[0]] @ w_l.t()
                z = bwd(fwd(z)) if not last else bwd(z)  # Z / dZ stored as bf16 (hidden); dZ rounded as a GEMM operand (last)
                agg = agg + _scatter_mean(z.index_select(0, ei[et][0]), ei[et][1], x[t].size(0))
            if last:
                pre = agg + bwd(zroot + bias32.to(dtype))  # root block of dZ (= output gradient) rounded as a GEMM operand
            else:
                pre = bwd(agg + zroot + bias32.to(dtype))  # G[l+1] stored as bf16
            if last:
                nxt[t] = pre
            else:
                h = F.relu(pre)
                p = float(ora.dropout)
                if training and p > 0:
                    h = dropout_fn(h, p, True, f"L{l}.{t}") if dropout_fn is not None else F.dropout(h, p, True)
                nxt[t] = fwd(h)  # hidden activations stored as bf16
        x = nxt
    logits = x["rooms"]
    y = batch["rooms"].y
    mask = y != 25
    loss = F.cross_entropy(logits[mask], y[mask])
    loss.backward()
    grads = {n: (t.grad.detach() if t.grad is not None else None) for n, t in leaves.items()}
    return logits.detach(), loss.detach(), grads
