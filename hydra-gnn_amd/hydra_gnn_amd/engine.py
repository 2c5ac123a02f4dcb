"""Host side of the native network executor (``hmp_net_*`` in ``include/hydra_mp.h``).

:class:`NativeNet` turns a model description (node/edge types, per-layer convs with their
``nn.Parameter`` objects) into an ``hmp_net_spec``, keeps all parameters in ONE flat fp32 buffer
(the named parameters are views into it, so ``state_dict()`` keeps the reference's PyG keys while
the optimiser / all-reduce see one contiguous tensor), owns the device workspace and exposes

* ``forward(data)`` -- autograd-compatible (one ``torch.autograd.Function`` around
  ``hmp_net_forward`` / ``hmp_net_backward``), so the reference's training loop
  (``loss.backward(); opt.step()``, ``src/hydra_gnn/base_training_job.py:202-216``) works unchanged;
* :class:`TrainStep` -- the same loop body as two native phases (fwd+loss+bwd, Adam) captured in
  hipGraphs, with one flat gradient all-reduce between them for data parallelism.

torch is used for device memory, streams and ``torch.distributed`` only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT_ELU, ACT_NONE, ACT_RELU, CONV_GAT, CONV_SAGE

EdgeType = Tuple[str, str, str]


class ConvDesc:
    """One conv of one layer: kind, edge type and its parameters (by role)."""

    def __init__(self, kind: int, edge_type: EdgeType, f_out: int, params: Dict[str, Optional[nn.Parameter]], **gat):
        self.kind, self.edge_type, self.f_out, self.params = kind, tuple(edge_type), f_out, params
        self.gat = gat  # heads, concat, self_loops, edge_dim, fill_mean, shared_lin
        self.active = True


class LayerDesc:
    def __init__(self, convs: List[ConvDesc], out_dims: Dict[str, int], act: int, dropout: float, group_mean: bool = False,
                 passthrough: Sequence[str] = ()):
        self.convs, self.out_dims, self.act, self.dropout, self.group_mean = convs, out_dims, act, dropout, group_mean
        # first layer only: node types whose input features stay visible to the next layer (`x_dict.update(pre_mp(..))`)
        self.passthrough = list(passthrough)


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.HydraMPError(
            f"{what} is on {t.device}: hydra_gnn_amd runs on MI355X (cuda/HIP device) only and has no CPU fallback"
        )


class _BatchHolder:
    """ctypes batch descriptor + the tensors it points at (kept alive while kernels run)."""

    def __init__(self):
        self.c = _lib.Batch()
        self.keep: List[torch.Tensor] = []
        self.n_nodes: List[int] = []
        self.n_edges: List[int] = []
        self.edge_tensors: List[torch.Tensor] = []
        # an input had to be copied (dtype / layout conversion): the descriptor points at a private copy, so it must not be
        # reused for a later step (in-place edits of the caller's tensor would be missed)
        self.converted = False


class NativeNet:
    def __init__(self, node_types: Sequence[str], in_dims: Dict[str, int], edge_types: Sequence[EdgeType],
                 layers: List[LayerDesc], readout: str, pool_edge_type: Optional[EdgeType] = None,
                 count_types: Sequence[str] = (), aux_readout: Optional[str] = None):
        self.node_types = list(node_types)
        self.edge_types = [tuple(e) for e in edge_types]
        self.in_dims = dict(in_dims)
        self.layers = layers
        self.readout = readout
        # second output node type of the two-headed task (heterogeneous_network.py:123-135): forward returns (readout, aux)
        self.aux_readout = aux_readout
        assert aux_readout is None or (aux_readout != readout and pool_edge_type is None)
        self.pool_edge_type = tuple(pool_edge_type) if pool_edge_type is not None else None
        # node types without features whose node COUNT matters (virtual pool targets)
        self.count_types = list(count_types)
        assert len(self.node_types) <= _lib.MAX_NODE_TYPES and len(self.edge_types) <= _lib.MAX_EDGE_TYPES
        # edge types whose convs read edge attributes (GAT_edge) -> attribute width
        self.edge_dims: Dict[EdgeType, int] = {}
        for layer in layers:
            for conv in layer.convs:
                d = int(conv.gat.get("edge_dim", 0) or 0)
                if d:
                    self.edge_dims[conv.edge_type] = d
        self._mark_liveness()
        self._layout_params()
        self._handle = None
        self._ws = None
        self._caps = None
        self._flat: Optional[torch.Tensor] = None
        self._lib = None
        self._fwd_token = 0
        self._compute_bf16 = False
        self._plan_key = None  # versions of the edge lists whose plan the workspace holds (predict() reuses it across frames)
        self._plan_tensors = None  # ... and the edge tensors themselves (identity is the test; holding them pins their addresses)
        self._agg_first: Dict[Tuple[int, int], bool] = {}  # (layer, conv) -> evaluated aggregate-first (decided with the first batch)

    def set_compute(self, precision: str) -> None:
        """'fp32' (default): every projection on the exact fp32 MFMA path.  'bf16': GEMM calls in the throughput-bound regime
        (>= 1024 64x64 output tiles, i.e. BASELINE config 5) round their operands to bf16 and accumulate in fp32; storage stays
        fp32.  An explicit precision choice: outside the 1e-5 parity bar of the fp32 path."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self._compute_bf16 = precision == "bf16"
        if self._handle is not None:
            _lib.check(self._lib.hmp_net_set_compute(self._handle, int(self._compute_bf16)))

    # ---- static analysis ---------------------------------------------------------------------
    def _mark_liveness(self) -> None:
        """A conv is live iff its output can reach the readout (reference computes the others and
        throws them away: heterogeneous_network.py:121-122).  Dead convs keep their parameters
        (state_dict compatible) but receive no gradient, exactly like ``grad is None`` in torch."""
        needed = {self.readout} | ({self.aux_readout} if self.aux_readout is not None else set())
        for layer in reversed(self.layers):
            nxt = set()
            for conv in layer.convs:
                s, _, t = conv.edge_type
                conv.active = t in needed
                if conv.active:
                    nxt.update((s, t))
            nxt.update(t for t in layer.passthrough if t in needed)
            needed = nxt

    def _layout_params(self) -> None:
        roles = ("w0", "w1", "w2", "a0", "a1", "a2", "b0")
        seen: Dict[int, int] = {}
        order: List[Tuple[nn.Parameter, bool]] = []
        for active_pass in (True, False):
            for layer in self.layers:
                for conv in layer.convs:
                    for r in roles:
                        p = conv.params.get(r)
                        if p is None or id(p) in seen:
                            continue
                        live = conv.active and self._role_used(conv, r)
                        if live != active_pass:
                            continue
                        seen[id(p)] = len(order)
                        order.append((p, live))
        # a parameter shared by several convs (GAT lin_src == lin_dst) is active if any user is
        off = 0
        self.param_offsets: Dict[int, int] = {}
        self.params: List[nn.Parameter] = []
        self.param_active: List[bool] = []
        for p, live in order:
            self.param_offsets[id(p)] = off
            self.params.append(p)
            self.param_active.append(live)
            off += ((p.numel() + 3) // 4) * 4  # keep every tensor 16-byte aligned in the flat buffer
            if live:
                self.n_active = off
        if not any(self.param_active):
            self.n_active = 0
        self.n_params = off

    @staticmethod
    def _role_used(conv: ConvDesc, role: str) -> bool:
        if conv.kind == CONV_SAGE:
            return role in ("w0", "b0", "w1")
        # GAT: lin_dst of a same-type conv built from a tuple is never used
        s, _, t = conv.edge_type
        if role == "w1" and s == t and not conv.gat.get("shared_lin", False):
            return False
        return True

    # ---- native objects ------------------------------------------------------------------------
    def _spec(self) -> _lib.NetSpec:
        sp = _lib.NetSpec()
        nt = {t: i for i, t in enumerate(self.node_types)}
        et = {e: i for i, e in enumerate(self.edge_types)}
        sp.n_node_types, sp.n_edge_types, sp.n_layers = len(self.node_types), len(self.edge_types), len(self.layers)
        for t, i in nt.items():
            sp.in_dim[i] = int(self.in_dims.get(t, 0))
        for e, i in et.items():
            sp.edge_src[i], sp.edge_dst[i] = nt[e[0]], nt[e[2]]
        sp.readout_type = nt[self.readout]
        sp.pool_edge_type = et[self.pool_edge_type] if self.pool_edge_type is not None else -1
        sp.aux_readout_type = nt[self.aux_readout] if self.aux_readout is not None else -1
        sp.n_params, sp.n_active_params = self.n_params, self.n_active
        for l, layer in enumerate(self.layers):
            ls = sp.layers[l]
            ls.n_convs, ls.act, ls.dropout, ls.group_mean = len(layer.convs), layer.act, float(layer.dropout), int(layer.group_mean)
            for t, i in nt.items():
                ls.out_dim[i] = int(layer.out_dims.get(t, 0))
                ls.passthrough[i] = int(t in layer.passthrough)
            for c, conv in enumerate(layer.convs):
                cs = ls.convs[c]
                cs.kind, cs.edge_type = conv.kind, et[conv.edge_type]
                cs.src, cs.dst, cs.f_out = nt[conv.edge_type[0]], nt[conv.edge_type[2]], conv.f_out
                cs.heads = int(conv.gat.get("heads", 1))
                cs.concat = int(conv.gat.get("concat", 0))
                cs.self_loops = int(conv.gat.get("self_loops", 0))
                cs.edge_dim = int(conv.gat.get("edge_dim", 0) or 0)
                cs.fill_mean = int(conv.gat.get("fill_mean", 0))
                cs.shared_lin = int(conv.gat.get("shared_lin", 0))
                cs.active = int(conv.active)
                cs.agg_first = int(self._agg_first.get((l, c), False))
                cs.att_dropout = float(conv.gat.get("dropout", 0.0) or 0.0)
                for r in ("w0", "w1", "w2", "a0", "a1", "a2", "b0"):
                    p = conv.params.get(r)
                    setattr(cs, r, self.param_offsets[id(p)] if p is not None else -1)
        return sp

    def _decide_agg_first(self, h: Optional["_BatchHolder"]) -> None:
        """Which SAGE convs are evaluated in the reference's own order -- mean of the source rows, then ``lin_l`` on the (few)
        destination rows ([PyG] SAGEConv) -- instead of projecting every source row first (both are exact).
        Decided ONCE, from the sizes of the first batch the handle is built for (the choice is part of the executor's static
        layout): a conv between two node types whose source type is large (>= 32 768 rows) and at least 16x the destination type,
        with at most two edges per source row -- objects -> rooms at 10^6 objects; at most one per source type and layer.
        ``HMP_AGG_FIRST=0`` never, ``=1`` every eligible conv whatever the sizes (tests)."""
        self._agg_first = {}
        mode = os.environ.get("HMP_AGG_FIRST")
        if mode == "0" or (h is None and mode != "1"):
            return
        nt = {t: i for i, t in enumerate(self.node_types)}
        et = {e: i for i, e in enumerate(self.edge_types)}
        for l, layer in enumerate(self.layers):
            taken = set()
            for c, conv in enumerate(layer.convs):
                s, _, t = conv.edge_type
                if conv.kind != CONV_SAGE or not conv.active or s == t or s in taken:
                    continue
                if mode != "1":
                    ns, nd, ne = h.n_nodes[nt[s]], h.n_nodes[nt[t]], h.n_edges[et[conv.edge_type]]
                    if not (ns >= 32768 and ns >= 16 * max(nd, 1) and 0 < ne <= 2 * ns):
                        continue
                self._agg_first[(l, c)] = True
                taken.add(s)

    def _ensure_handle(self, h: Optional["_BatchHolder"] = None):
        if self._handle is None:
            self._lib = _lib.require_device()
            self._decide_agg_first(h)
            h = C.c_void_p()
            sp = self._spec()
            _lib.check(self._lib.hmp_net_create(C.byref(sp), C.byref(h)))
            self._handle = h
            if self._compute_bf16:
                _lib.check(self._lib.hmp_net_set_compute(h, 1))
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None and self._lib is not None:
                self._lib.hmp_net_destroy(self._handle)
        except Exception:
            pass

    # ---- flat parameters -------------------------------------------------------------------------
    def flat_params(self, full_check: bool = True) -> torch.Tensor:
        """The flat fp32 buffer the named parameters are views of (re-created after ``.to()``)."""
        p0 = self.params[0]
        _require_cuda(p0.data, "model parameters")
        ok = self._flat is not None and self._flat.device == p0.device
        if ok:
            base = self._flat.data_ptr()
            # `.to()` / `.float()` re-home every parameter; the first and the last one are checked on the hot path (one call
            # per step), all of them when the check is forced
            probe = self.params if full_check else (self.params[0], self.params[-1])
            for p in probe:
                if p.data.data_ptr() != base + 4 * self.param_offsets[id(p)] or p.dtype != torch.float32:
                    ok = False
                    break
        if not ok:
            flat = torch.zeros(self.n_params + 4, dtype=torch.float32, device=p0.device)
            for p in self.params:
                if p.dtype != torch.float32:
                    raise _lib.HydraMPError("hydra_gnn_amd computes in fp32: parameter dtype must be float32")
                off = self.param_offsets[id(p)]
                view = flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
            self._flat = flat
        return self._flat

    # ---- batch + workspace -------------------------------------------------------------------------
    def make_batch(self, data, labels: Optional[torch.Tensor] = None) -> _BatchHolder:
        h = _BatchHolder()
        x_dict = data.x_dict
        ei_dict = data.edge_index_dict
        for i, t in enumerate(self.node_types):
            if t in x_dict and self.in_dims.get(t, 0) > 0:
                x = x_dict[t]
                _require_cuda(x, f"x_dict['{t}']")
                if x.dtype != torch.float32 or x.stride(-1) != 1 or (x.dim() == 2 and x.size(0) > 1 and x.stride(0) < x.size(1)):
                    x = x.to(torch.float32).contiguous()
                    h.converted = True
                if x.size(1) != self.in_dims[t]:
                    raise _lib.HydraMPError(f"x_dict['{t}'] has {x.size(1)} features, model expects {self.in_dims[t]}")
                h.keep.append(x)
                h.c.n_nodes[i] = x.size(0)
                h.c.d_x[i] = x.data_ptr()
                h.c.ldx[i] = x.stride(0) if x.size(0) > 1 else x.size(1)
            else:
                h.c.n_nodes[i] = int(data[t].num_nodes) if t in self.count_types or t in x_dict else 0
            h.n_nodes.append(int(h.c.n_nodes[i]))
        for i, e in enumerate(self.edge_types):
            if e in ei_dict:
                ei = ei_dict[e]
                _require_cuda(ei, f"edge_index_dict[{e}]")
                if ei.dtype != torch.int64 or not ei.is_contiguous():
                    ei = ei.to(torch.int64).contiguous()
                    h.converted = True
                h.keep.append(ei)
                h.edge_tensors.append(ei)
                h.c.n_edges[i] = ei.size(1)
                h.c.d_edge_index[i] = ei.data_ptr() if ei.size(1) > 0 else None
            else:
                # PyG's HeteroConv SKIPS a conv whose edge type is absent from the dict (no root / bias term, smaller group-mean
                # divisor), which differs from a present-but-empty `[2, 0]` edge_index.  The reference's loaders always
                # materialise every type (`fill_missing_edge_index`, mp3d_dataset.py:220-255); the executor's stacked operands
                # are built for the full conv set, so an absent key is refused instead of silently computing something else.
                raise _lib.HydraMPError(
                    f"edge_index_dict has no entry for {e}: pass an empty [2, 0] int64 edge_index for edge types without "
                    "edges (the reference does: fill_missing_edge_index, mp3d_dataset.py:220-255)")
            h.n_edges.append(int(h.c.n_edges[i]))
        if self.edge_dims:
            ea_dict = data.edge_attr_dict
            for i, e in enumerate(self.edge_types):
                d = self.edge_dims.get(e, 0)
                if d == 0 or h.n_edges[i] == 0:
                    continue
                if e not in ea_dict:
                    raise _lib.HydraMPError(f"edge_attr_dict[{e}] is required by the GAT_edge convs")
                ea = ea_dict[e]
                _require_cuda(ea, f"edge_attr_dict[{e}]")
                if ea.dtype != torch.float32 or not ea.is_contiguous():
                    ea = ea.to(torch.float32).contiguous()
                    h.converted = True
                if ea.dim() != 2 or ea.size(0) != h.n_edges[i] or ea.size(1) != d:
                    raise _lib.HydraMPError(f"edge_attr_dict[{e}] must be [{h.n_edges[i]}, {d}], got {tuple(ea.shape)}")
                h.keep.append(ea)
                h.c.d_edge_attr[i] = ea.data_ptr()
        nt = {t: i for i, t in enumerate(self.node_types)}
        out_type = self.pool_edge_type[2] if self.pool_edge_type is not None else self.readout
        h.c.n_out = h.n_nodes[nt[out_type]]
        # the batch as a union of graphs ([PyG] Batch.ptr per node store + the collation's host-side maximum): optional
        mg = int(getattr(data, "max_graph_nodes", 0) or 0)
        ng = 0
        if mg > 0:
            for i, t in enumerate(self.node_types):
                st_ = data[t] if t in x_dict else None
                p = getattr(st_, "ptr", None) if st_ is not None else None
                if p is None or not p.is_cuda or p.dtype != torch.int64 or not p.is_contiguous() or int(p[-1:].numel()) == 0:
                    if h.n_nodes[i] > 0:
                        ng = 0
                        break
                    continue
                if ng and p.numel() - 1 != ng:
                    ng = 0
                    break
                ng = p.numel() - 1
                h.keep.append(p)
                h.c.d_node_ptr[i] = p.data_ptr()
        h.c.n_graphs = ng
        h.c.max_graph_nodes = mg if ng else 0
        if ng:  # graph-sorted edge lists (collate's per-edge-type ptr)
            for i, e in enumerate(self.edge_types):
                st_ = data[e] if e in data.edge_types else None
                p = getattr(st_, "ptr", None) if st_ is not None else None
                # `ptr` vouches for ONE edge_index (data.collate stamps its version): an edge list edited in place since then is
                # described without it (any edge order accepted; a replaced edge_index dropped `ptr` already)
                pv = getattr(st_, "ptr_version", None) if st_ is not None else None
                if pv is not None and int(h.edge_tensors[i]._version) != int(pv):
                    p = None
                if p is not None and p.is_cuda and p.dtype == torch.int64 and p.is_contiguous() and p.numel() == ng + 1:
                    h.keep.append(p)
                    h.c.d_edge_ptr[i] = p.data_ptr()
        if labels is not None:
            _require_cuda(labels, "labels")
            lab = labels.to(torch.int64).contiguous()
            if lab is not labels:
                h.converted = True
            if lab.numel() != h.c.n_out:
                raise _lib.HydraMPError(f"{lab.numel()} labels for {h.c.n_out} output rows")
            h.keep.append(lab)
            h.c.d_labels = lab.data_ptr()
        return h

    def _ensure_workspace(self, h: _BatchHolder, device) -> None:
        handle = self._ensure_handle(h)
        need_n, need_e = h.n_nodes, h.n_edges
        if self._caps is not None and self._ws is not None and self._ws.device == device:
            cn, ce = self._caps
            if all(a <= b for a, b in zip(need_n, cn)) and all(a <= b for a, b in zip(need_e, ce)):
                return
            need_n = [max(a, b) for a, b in zip(need_n, cn)]
            need_e = [max(a, b) for a, b in zip(need_e, ce)]
        cn = [max(1, int(v)) for v in need_n]
        ce = [max(1, int(v)) for v in need_e]
        cn_c = (C.c_int32 * len(cn))(*cn)
        ce_c = (C.c_int64 * len(ce))(*ce)
        nbytes = int(self._lib.hmp_net_workspace_bytes(handle, cn_c, ce_c))
        torch.cuda.synchronize(device)
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _lib.check(self._lib.hmp_net_bind_workspace(handle, self._ws.data_ptr(), nbytes, cn_c, ce_c))
        self._caps = (cn, ce)
        self._fwd_token += 1  # activations of an earlier forward are gone: its backward must fail loudly

    def _ws_view(self, ptr: int, rows: int, ld: int) -> torch.Tensor:
        off = ptr - self._ws.data_ptr()
        assert 0 <= off and off + rows * ld * 4 <= self._ws.numel() and off % 4 == 0
        return self._ws[off:off + rows * ld * 4].view(torch.float32).view(rows, ld)

    # ---- autograd-compatible forward -------------------------------------------------------------
    def forward(self, data, training: bool, seed: int = 0, rng_step: int = 0) -> torch.Tensor:
        flat = self.flat_params()
        h = self.make_batch(data)
        with torch.cuda.device(flat.device):
            self._ensure_workspace(h, flat.device)
            return _NetFunction.apply(self, h, bool(training), int(seed), int(rng_step), *self.params)

    # ---- inference hot loop (bin/room_classification_server:273-299) -------------------------------------------------
    def predict(self, data, n_classes: int) -> torch.Tensor:
        """``forward(data).argmax(dim=1).cpu()`` of the reference's ``GnnModel.infer`` without autograd, without copying the
        logits and with ONE small D2H: eval-mode native forward, ``hmp_argmax_rows`` on the executor's output, labels into a
        pinned host buffer.  Returns a host int64 tensor (a view of that buffer: valid until the next call)."""
        flat = self.flat_params(full_check=False)
        h = self.make_batch(data)
        dev = flat.device
        n = int(h.c.n_out)
        with torch.cuda.device(dev):
            self._ensure_workspace(h, dev)
            if getattr(self, "_pred_dev", None) is None or self._pred_dev.numel() < n or self._pred_dev.device != dev:
                cap = max(256, 2 * n)
                self._pred_dev = torch.empty(cap, dtype=torch.int64, device=dev)
                self._pred_host = torch.empty(cap, dtype=torch.int64).pin_memory()
                self._pred_done = torch.cuda.Event()
            out_p, ld = C.c_void_p(), C.c_int32()
            st = _lib.stream_ptr()
            # consecutive frames with the SAME edge tensors (unchanged in place): the plan of the previous call is reused
            # (bin/room_classification_server:273-299 re-infers on a graph whose topology did not change)
            # "same" = the very same tensor OBJECTS as the previous call (kept alive in `_plan_tensors`, so the allocator cannot
            # hand their addresses to a new frame's edge lists) at the same `_version`: a fresh tensor that merely lands on a
            # recycled address rebuilds the plan
            key = None if h.converted else (id(self._ws), tuple(h.n_nodes), tuple(t._version for t in h.edge_tensors))
            prev = self._plan_tensors
            same = (key is not None and key == self._plan_key and prev is not None and len(prev) == len(h.edge_tensors)
                    and all(a is b for a, b in zip(prev, h.edge_tensors)))
            h.c.plan_valid = 1 if same else 0
            _lib.check(self._lib.hmp_net_forward(self._handle, C.byref(h.c), flat.data_ptr(), 0, 0, 0, C.byref(out_p), C.byref(ld), st))
            self._fwd_token += 1
            self._plan_key = key
            self._plan_tensors = list(h.edge_tensors) if key is not None else None
            if n > 0:
                _lib.check(self._lib.hmp_argmax_rows(out_p.value, ld.value, n, int(n_classes), self._pred_dev.data_ptr(), st))
                self._pred_host[:n].copy_(self._pred_dev[:n], non_blocking=True)
            self._pred_done.record()
            self._pred_done.synchronize()
        return self._pred_host[:n]

    def _forward_raw(self, h: _BatchHolder, training: bool, seed: int, rng_step: int) -> torch.Tensor:
        out_p, ld = C.c_void_p(), C.c_int32()
        _lib.check(self._lib.hmp_net_forward(self._handle, C.byref(h.c), self._flat.data_ptr(), int(training), seed, rng_step,
                                             C.byref(out_p), C.byref(ld), _lib.stream_ptr()))
        self._fwd_token += 1
        self._plan_key = self._plan_tensors = None
        out = self._ws_view(out_p.value, int(h.c.n_out), ld.value).clone()
        if self.aux_readout is None:
            return out
        rows = C.c_int32()
        _lib.check(self._lib.hmp_net_aux_output(self._handle, C.byref(out_p), C.byref(ld), C.byref(rows)))
        return out, self._ws_view(out_p.value, rows.value, ld.value).clone()

    def _backward_raw(self, gout: Optional[torch.Tensor], gaux: Optional[torch.Tensor] = None) -> torch.Tensor:
        dev = (gout if gout is not None else gaux).device
        grads = torch.zeros(self.n_params + 4, dtype=torch.float32, device=dev)
        if self.aux_readout is None:
            _lib.check(self._lib.hmp_net_backward(self._handle, gout.data_ptr(), gout.stride(0), self._flat.data_ptr(),
                                                  grads.data_ptr(), None, _lib.stream_ptr()))
        else:
            _lib.check(self._lib.hmp_net_backward2(self._handle, gout.data_ptr() if gout is not None else None,
                                                   gout.stride(0) if gout is not None else 0,
                                                   gaux.data_ptr() if gaux is not None else None,
                                                   gaux.stride(0) if gaux is not None else 0, self._flat.data_ptr(),
                                                   grads.data_ptr(), None, _lib.stream_ptr()))
        return grads

    def hidden(self, layer: int, node_type: str) -> torch.Tensor:
        """Copy (fp32) of the stored output of ``layer`` (1-based) for ``node_type`` from the last forward (``hmp_net_hidden``;
        diagnosis / tests).  Dropped elements are ``-0.0``."""
        p, ld, rows, width, b16 = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self._lib.hmp_net_hidden(self._handle, int(layer), self.node_types.index(node_type), C.byref(p), C.byref(ld),
                                            C.byref(rows), C.byref(width), C.byref(b16)))
        esz = 2 if b16.value else 4
        off = p.value - self._ws.data_ptr()
        raw = self._ws[off:off + rows.value * ld.value * esz].view(torch.bfloat16 if b16.value else torch.float32)
        return raw.view(rows.value, ld.value)[:, :width.value].to(torch.float32).clone()

    STATUS_BITS = {1: "an edge endpoint is out of range (the edge was dropped)", 2: "a label is out of range",
                   4: "an edge lies outside the rows of the graph its Batch.ptr slice belongs to (stale or wrong edge `ptr`)"}

    def check_status(self) -> None:
        """Raise if a kernel flagged bad input data since the net was created (sticky device status word; synchronises).  The
        training loop reads the loss every step (``base_training_job.py:216``), which is where :meth:`TrainStep.loss` calls this."""
        _, status = self.read_state()
        if status:
            what = "; ".join(msg for bit, msg in self.STATUS_BITS.items() if status & bit) or "unknown bit"
            raise _lib.HydraMPError(f"device status word = {status}: {what}")

    def read_state(self) -> Tuple[int, int]:
        step, status = C.c_int32(), C.c_int32()
        _lib.check(self._lib.hmp_net_read_state(self._handle, C.byref(step), C.byref(status), _lib.stream_ptr()))
        return step.value, status.value


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: NativeNet, holder: _BatchHolder, training: bool, seed: int, rng_step: int, *params):
        out = net._forward_raw(holder, training, seed, rng_step)
        ctx.net, ctx.holder, ctx.token = net, holder, net._fwd_token
        return out

    @staticmethod
    def backward(ctx, gout, gaux=None):
        net = ctx.net
        if ctx.token != net._fwd_token:
            raise _lib.HydraMPError(
                "backward() after another forward() of the same model: the native executor keeps the "
                "activations of the LAST forward only (one forward, then its backward)"
            )
        if gout is not None:
            gout = gout.contiguous()
            if gout.stride(0) % 4 != 0 or gout.data_ptr() % 16 != 0:
                padded = torch.zeros(gout.size(0), ((gout.size(1) + 3) // 4) * 4, dtype=gout.dtype, device=gout.device)
                padded[:, : gout.size(1)] = gout
                gout = padded
        if gaux is not None:
            gaux = gaux.contiguous()  # staged by the executor (any row pitch)
        if gout is None and gaux is None:
            return (None,) * (5 + len(net.params))
        flat_g = net._backward_raw(gout, gaux)
        grads = []
        for p, live in zip(net.params, net.param_active):
            if not live:
                grads.append(None)
                continue
            off = net.param_offsets[id(p)]
            grads.append(flat_g[off:off + p.numel()].view(p.shape))
        return (None, None, None, None, None, *grads)


class TrainStep:
    """The loop body of ``BaseTrainingJob.train`` (``base_training_job.py:202-216``) as native code:
    phase A = plan + forward + masked CE + backward, [all-reduce], phase B = Adam.

    Gradients are SUMS over valid labels; ``{loss_sum, count}`` ride in the tail of the same flat
    buffer, so one all-reduce(sum) of ``n_active + 2`` floats makes the N-rank update equal to the
    single-process full-batch update (count-weighted mean), and every rank applies the identical Adam.
    """

    def __init__(self, net: NativeNet, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 ignored_label: int = 25, seed: int = 0, training: bool = True, use_graph: bool = True,
                 process_group=None, force_collective: bool = False, view=None, comm=None, broadcast_init: bool = True):
        self.net = net
        # comm: parallel.NativeComm -- the all-reduce is enqueued by the library on the executor's own stream (RCCL);
        # process_group (True = default group): the same collective through torch.distributed (its own stream + event waits)
        self.comm = comm
        # homogeneous models: callable that presents their `Data` through the hetero accessors the executor reads
        self.view = view
        # run the data-parallel launch structure (two graphs around one all-reduce) even with a single rank
        self.force_collective = bool(force_collective)
        self.args = _lib.TrainArgs(lr, betas[0], betas[1], eps, weight_decay, ignored_label, seed, int(training))
        self.use_graph = use_graph
        self.pg = process_group
        self._graphs = None
        self._key = None
        self._stream = None
        self.flat = net.flat_params()
        dev = self.flat.device
        n = net.n_params + 4
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        # Adam's t (and the dropout draw number) belong to the optimiser state, like m / v (torch.optim.Adam keeps
        # state['step'] per optimiser): a second TrainStep on the same net starts at t = 1, and a workspace re-bind
        # (a larger batch arrived) leaves it alone
        self.step_ctr = torch.zeros(4, dtype=torch.int32, device=dev)
        self.args.d_step = self.step_ctr.data_ptr()
        self._holder = None
        self._batch_key = None
        self._data_ref = None
        if broadcast_init and self._world() > 1:
            # every rank starts from rank 0's weights: identical seeds are not relied upon
            if self.comm is not None:
                self.comm.broadcast_(self.flat, net.n_params)
            else:
                from . import parallel

                parallel.broadcast_parameters(self.flat, group=None if self.pg is True else self.pg)

    def _world(self) -> int:
        if self.comm is not None:
            return self.comm.world
        if self.pg is None:
            return 1
        import torch.distributed as dist

        return dist.get_world_size(self.pg if self.pg is not True else None)

    def _all_reduce(self) -> None:
        if self.comm is not None:
            if self.comm.world > 1 or self.force_collective:
                self.comm.all_reduce_sum_(self.grads, self.net.n_active + 2)
            return
        if self._world() > 1 or self.force_collective:
            from . import parallel

            parallel.allreduce_flat(self.grads, self.net.n_active, group=None if self.pg is True else self.pg,
                                    force=self.force_collective)

    def _phase_a(self, h, st):
        net = self.net
        _lib.check(net._lib.hmp_net_step_fwd_bwd(net._handle, C.byref(h.c), self.flat.data_ptr(), self.grads.data_ptr(),
                                                 C.byref(self.args), st))

    def _phase_ab(self, h, st):
        """single rank: both phases in one native call (Adam may ride in the gradient un-pack kernel)"""
        net = self.net
        _lib.check(net._lib.hmp_net_step_fused(net._handle, C.byref(h.c), self.flat.data_ptr(), self.grads.data_ptr(),
                                               self.m.data_ptr(), self.v.data_ptr(), C.byref(self.args), st))

    def _phase_b(self, st):
        net = self.net
        _lib.check(net._lib.hmp_net_step_adam(net._handle, self.flat.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                                              self.v.data_ptr(), C.byref(self.args), st))

    def __call__(self, data, labels: torch.Tensor) -> None:
        net = self.net
        if net.flat_params(full_check=False) is not self.flat:
            raise _lib.HydraMPError("model parameters were moved after TrainStep was created")
        # the batch descriptor of an unchanged batch object is reused (hydra_gnn_amd.data.HeteroData stamps every mutation;
        # foreign containers are described afresh every step)
        stamp = getattr(data, "_mutation_stamp", None)
        key = (id(data), stamp(), id(labels), labels._version) if stamp is not None else None
        if key is not None and key == self._batch_key:
            h = self._holder
        else:
            h = net.make_batch(self.view(data) if self.view is not None else data, labels)
            self._batch_key = key if not h.converted else None
            self._data_ref = (data, labels)  # keeps id() unique while the key is live
        dev = self.flat.device
        with torch.cuda.device(dev):
            net._ensure_workspace(h, dev)
            net._fwd_token += 1  # the step overwrites the activations of any earlier forward()
            net._plan_key = net._plan_tensors = None
            if not self.use_graph:
                st = _lib.stream_ptr()
                if self._world() == 1 and not self.force_collective:
                    self._phase_ab(h, st)
                else:
                    self._phase_a(h, st)
                    self._all_reduce()
                    self._phase_b(st)
                self._holder = h
                return
            key = (tuple(h.n_nodes), tuple(h.n_edges), tuple(t.data_ptr() for t in h.keep), id(net._ws))
            if self._graphs is None or key != self._key:
                self._capture(h, key)
            ga, gb = self._graphs
            cur = torch.cuda.current_stream()
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                st = _lib.stream_ptr()
                _lib.check(net._lib.hmp_graph_launch(ga, st))
                if gb is not None:  # data parallel: the all-reduce sits between the two captured phases
                    self._all_reduce()
                    _lib.check(net._lib.hmp_graph_launch(gb, st))
            cur.wait_stream(self._stream)

    def _capture(self, h, key) -> None:
        net = self.net
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.flat.device)
        self._destroy_graphs()
        torch.cuda.synchronize(self.flat.device)
        with torch.cuda.stream(self._stream):
            st = _lib.stream_ptr()
            ga, gb = C.c_void_p(), C.c_void_p()
            single = self._world() == 1 and not self.force_collective  # no collective between the phases: ONE graph per step
            _lib.check(net._lib.hmp_graph_begin(st))
            try:
                if single:
                    self._phase_ab(h, st)
                else:
                    self._phase_a(h, st)
            finally:
                _lib.check(net._lib.hmp_graph_end(st, C.byref(ga)))
            if single:
                gb = None
            else:
                _lib.check(net._lib.hmp_graph_begin(st))
                try:
                    self._phase_b(st)
                finally:
                    _lib.check(net._lib.hmp_graph_end(st, C.byref(gb)))
        self._graphs, self._key, self._holder = (ga, gb), key, h

    def _destroy_graphs(self) -> None:
        if self._graphs is not None:
            for g in self._graphs:
                if g is not None:
                    self.net._lib.hmp_graph_destroy(g)
            self._graphs = None

    def __del__(self):
        try:
            self._destroy_graphs()
        except Exception:
            pass

    def run(self, holder: _BatchHolder) -> None:
        """One training step on a batch that is already described (``store.BatchStream.next``): no per-tensor Python, no
        descriptor cache -- the path a data loader that lives on the device drives.  Eager launches."""
        net = self.net
        if self.use_graph:
            raise _lib.HydraMPError("TrainStep.run steps a NEW batch every call: create the step with use_graph=False")
        net._fwd_token += 1
        net._plan_key = net._plan_tensors = None
        self._batch_key = None
        self._holder = holder
        st = _lib.stream_ptr()
        if self._world() == 1 and not self.force_collective:
            self._phase_ab(holder, st)
        else:
            self._phase_a(holder, st)
            self._all_reduce()
            self._phase_b(st)

    def set_lr(self, lr: float) -> None:
        """Follow a learning-rate scheduler (``StepLR`` in ``base_training_job.py:186-188``): takes effect with the next
        step.  Captured graphs carry the rate as a kernel argument and are re-captured."""
        if C.c_float(float(lr)).value != self.args.lr:
            self.args.lr = float(lr)
            self._destroy_graphs()
            self._key = None

    def steps_taken(self) -> int:
        """Adam's t after the last step (synchronises)."""
        return int(self.step_ctr[0].item())

    def loss(self) -> float:
        """mean CE over the valid labels of the last step (global when data parallel); synchronises."""
        t = self.grads[self.net.n_active: self.net.n_active + 2].tolist()
        self.net.check_status()
        return t[0] / max(t[1], 1.0)
