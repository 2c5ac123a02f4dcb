"""Device-resident dataset + device-side collation.

The reference feeds the training loop through PyG's ``DataLoader`` (``base_training_job.py:164-178``): every step it
collates ``batch_size`` ``HeteroData`` objects on the host (``Batch.from_data_list``) and copies the result to the device
(``batch.to(device)``, :205).  With the native step at ~0.1 ms that host work dominates.  :class:`GraphStore` uploads the
whole list of graphs ONCE as packed per-type arrays (MP3D: 90 scenes x 5 trajectories of 10^2-node graphs -- a few hundred
MB) and :meth:`GraphStore.collate` assembles a batch with the gather kernels of ``csrc/collate.hip``
(``hmp_collate_rows`` / ``hmp_collate_edges``).  The host only computes the ``[B + 1]`` offset vectors of the batch (numpy
cumsums over per-graph counts it keeps) and ships them in ONE small pinned H2D copy.

The result is a :class:`hydra_gnn_amd.data.HeteroData` on the device with the same content, order and dtypes as
``data.collate(graphs).to(device)`` (bit-identical: ``tests/test_gpu_collate.py``), incl. ``batch`` / ``ptr`` vectors and
``num_graphs``.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from . import _lib
from .data import EdgeType, HeteroData


class _Packed:
    """all graphs' rows of one attribute back to back + [G + 1] row offsets (host copy kept for the offset arithmetic)"""

    def __init__(self, parts: List[torch.Tensor], device):
        self.counts = np.array([int(p.size(0)) for p in parts], dtype=np.int64)
        self.ptr_host = np.concatenate([[0], np.cumsum(self.counts)]).astype(np.int64)
        self.data = torch.cat(parts, dim=0).contiguous().to(device)
        self.ptr = torch.from_numpy(self.ptr_host).to(device)
        self.row_shape = tuple(self.data.shape[1:])
        self.row_bytes = int(self.data.element_size()) * (int(np.prod(self.row_shape, dtype=np.int64)) if self.row_shape else 1)
        if self.row_bytes % 4 != 0:
            raise _lib.HydraMPError(f"attribute rows of {self.row_bytes} bytes: the collate kernels move 4-byte units "
                                    "(store bool / int8 masks as int32)")


class GraphStore:
    def __init__(self, graphs: Sequence[HeteroData], device="cuda:0"):
        assert len(graphs) > 0
        self.device = torch.device(device)
        self.lib = _lib.require_device()
        self.n_graphs = len(graphs)
        g0 = graphs[0]
        self.node_types = list(g0.node_types)
        self.edge_types: List[EdgeType] = list(g0.edge_types)
        # node attributes (tensors whose first dimension is the node count)
        self.node_attrs: Dict[str, Dict[str, _Packed]] = {}
        self.node_counts: Dict[str, np.ndarray] = {}
        self.count_only: Dict[str, bool] = {}
        for t in self.node_types:
            keys = [k for k in g0[t].keys() if isinstance(getattr(g0[t], k), torch.Tensor)]
            self.node_attrs[t] = {}
            for k in keys:
                self.node_attrs[t][k] = _Packed([getattr(g[t], k) for g in graphs], self.device)
            self.node_counts[t] = np.array([int(g[t].num_nodes) for g in graphs], dtype=np.int64)
            self.count_only[t] = "num_nodes" in g0[t] and "x" not in g0[t]
        # edges: graph-local indices, [2][E_total]
        self.edge_index: Dict[EdgeType, torch.Tensor] = {}
        self.edge_ptr: Dict[EdgeType, torch.Tensor] = {}
        self.edge_ptr_host: Dict[EdgeType, np.ndarray] = {}
        self.edge_attr: Dict[EdgeType, _Packed] = {}
        for e in self.edge_types:
            eis = [g[e].edge_index.to(torch.int64) for g in graphs]
            counts = np.array([int(ei.size(1)) for ei in eis], dtype=np.int64)
            ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            self.edge_ptr_host[e] = ptr
            self.edge_ptr[e] = torch.from_numpy(ptr).to(self.device)
            self.edge_index[e] = torch.cat(eis, dim=1).contiguous().to(self.device)
            if "edge_attr" in g0[e]:
                self.edge_attr[e] = _Packed([g[e].edge_attr for g in graphs], self.device)
        # staging buffer for the per-batch offset vectors (pinned: the H2D copy is asynchronous)
        self._stage = None
        self._stage_dev = None
        self._copied = None  # event: the previous batch's offset copy has left the pinned buffer

    # ---------------------------------------------------------------------------------------------------------------
    def collate(self, ids: Sequence[int]) -> HeteroData:
        """Batch of graphs ``ids`` (in that order) on the device; equals ``data.collate([graphs[i] for i in ids]).to(device)``."""
        sel_host = np.asarray(ids, dtype=np.int64)
        B = int(sel_host.size)
        assert B > 0 and sel_host.min() >= 0 and sel_host.max() < self.n_graphs
        st = _lib.stream_ptr()
        # ---- host: offsets of every node / edge type in the batch, packed into one int64 staging vector
        node_off: Dict[str, np.ndarray] = {}
        for t in self.node_types:
            node_off[t] = np.concatenate([[0], np.cumsum(self.node_counts[t][sel_host])]).astype(np.int64)
        edge_off: Dict[EdgeType, np.ndarray] = {}
        for e in self.edge_types:
            p = self.edge_ptr_host[e]
            edge_off[e] = np.concatenate([[0], np.cumsum(p[sel_host + 1] - p[sel_host])]).astype(np.int64)
        n_vec = len(self.node_types) + len(self.edge_types)
        words = n_vec * (B + 1) + (B + 1) // 2 + 1  # + sel as int32
        if self._copied is not None:
            self._copied.synchronize()  # the pinned buffer is about to be rewritten
        if self._stage is None or self._stage.numel() < words:
            self._stage = torch.empty(max(words, 4096), dtype=torch.int64).pin_memory()
            self._stage_dev = torch.empty_like(self._stage, device=self.device)
            self._copied = torch.cuda.Event()
        stage = self._stage.numpy()
        pos, where = 0, {}
        for key, vec in list(node_off.items()) + list(edge_off.items()):
            stage[pos:pos + B + 1] = vec
            where[key] = pos
            pos += B + 1
        sel32 = stage[pos:pos + (B + 1) // 2 + 1].view(np.int32)
        sel32[:B] = sel_host.astype(np.int32)
        sel_pos = pos
        self._stage_dev[:words].copy_(self._stage[:words], non_blocking=True)
        self._copied.record()
        dev = self._stage_dev
        sel_ptr = dev[sel_pos:].data_ptr()
        off_ptr = lambda key: dev[where[key]:].data_ptr()

        out = HeteroData()
        out.num_graphs = B
        lib = self.lib
        for t in self.node_types:
            n_out = int(node_off[t][-1])
            for k, pk in self.node_attrs[t].items():
                dst = torch.empty((n_out,) + pk.row_shape, dtype=pk.data.dtype, device=self.device)
                _lib.check(lib.hmp_collate_rows(pk.data.data_ptr(), pk.row_bytes, pk.ptr.data_ptr(), sel_ptr, off_ptr(t), B, n_out,
                                                dst.data_ptr(), st))
                setattr(out[t], k, dst)
            if self.count_only[t]:
                out[t].num_nodes = n_out
            ptr = dev[where[t]:where[t] + B + 1].clone()
            out[t].ptr = ptr
            out[t].batch = torch.repeat_interleave(torch.arange(B, dtype=torch.int64, device=self.device), ptr[1:] - ptr[:-1],
                                                   output_size=n_out)
        for e in self.edge_types:
            e_out = int(edge_off[e][-1])
            dst = torch.empty((2, e_out), dtype=torch.int64, device=self.device)
            src = self.edge_index[e]
            _lib.check(lib.hmp_collate_edges(src.data_ptr(), int(src.size(1)), self.edge_ptr[e].data_ptr(), sel_ptr, off_ptr(e),
                                             off_ptr(e[0]), off_ptr(e[2]), B, e_out, dst.data_ptr(), st))
            out[e].edge_index = dst
            if e in self.edge_attr:
                pk = self.edge_attr[e]
                ea = torch.empty((e_out,) + pk.row_shape, dtype=pk.data.dtype, device=self.device)
                _lib.check(lib.hmp_collate_rows(pk.data.data_ptr(), pk.row_bytes, self.edge_ptr[e].data_ptr(), sel_ptr, off_ptr(e), B, e_out,
                                                ea.data_ptr(), st))
                out[e].edge_attr = ea
        return out


class BatchStream:
    """A DataLoader's inner loop for ONE network, without the DataLoader: every :meth:`next` assembles a fresh batch of
    ``batch_size`` graphs on the device (``hmp_collator_run``: one host call, one kernel, outputs in buffers allocated once) and
    hands back the executor's batch descriptor ready made -- no per-tensor Python between two training steps
    (``base_training_job.py:202-216`` spends it in ``DataLoader.__next__`` + ``batch.to(device)`` + ``make_batch``).

        stream = store.stream(model, batch_size=32, label_type="rooms")
        for ids in sampler:                  # e.g. a shuffled permutation cut into batches
            step.run(stream.next(ids))       # TrainStep.run: fwd + loss + bwd + Adam on that batch

    The buffers are reused by the next call: a batch is valid until then (same-stream ordering makes that safe for everything
    already enqueued).  ``stream.data()`` presents the current batch as a ``HeteroData`` of views for code that wants one."""

    def __init__(self, store: "GraphStore", net, batch_size: int, label_type: str, label_key: str = "y"):
        import ctypes as C

        self.store, self.B = store, int(batch_size)
        self.net = net
        lib = store.lib
        dev = store.device
        slots: List[object] = list(store.node_types) + list(store.edge_types)
        slot_of = {k: i for i, k in enumerate(slots)}
        host_ptrs = []
        for t in store.node_types:
            host_ptrs.append(np.concatenate([[0], np.cumsum(store.node_counts[t])]).astype(np.int64))
        for e in store.edge_types:
            host_ptrs.append(np.ascontiguousarray(store.edge_ptr_host[e], dtype=np.int64))
        self._host_ptrs = host_ptrs
        self._node_ptr_dev = {t: torch.from_numpy(host_ptrs[slot_of[t]]).to(dev) for t in store.node_types}
        # capacity of a batch per slot: B times the largest graph (ids may repeat)
        self.cap = [int(np.diff(p).max()) * self.B for p in host_ptrs]
        items, self._bufs, self._what = [], [], []
        for t in store.node_types:
            for k, pk in store.node_attrs[t].items():
                it = _lib.CollateItem(pk.data.data_ptr(), self._node_ptr_dev[t].data_ptr(), 0, pk.row_bytes, slot_of[t], 0, 0)
                items.append(it)
                self._bufs.append(torch.empty((max(self.cap[slot_of[t]], 1),) + pk.row_shape, dtype=pk.data.dtype, device=dev))
                self._what.append(("node", t, k))
        for e in store.edge_types:
            src = store.edge_index[e]
            it = _lib.CollateItem(src.data_ptr(), store.edge_ptr[e].data_ptr(), int(src.size(1)), 0, slot_of[e], slot_of[e[0]], slot_of[e[2]])
            items.append(it)
            self._bufs.append(torch.empty(2 * max(self.cap[slot_of[e]], 1), dtype=torch.int64, device=dev))
            self._what.append(("edge", e, "edge_index"))
            if e in store.edge_attr:
                pk = store.edge_attr[e]
                items.append(_lib.CollateItem(pk.data.data_ptr(), store.edge_ptr[e].data_ptr(), 0, pk.row_bytes, slot_of[e], 0, 0))
                self._bufs.append(torch.empty((max(self.cap[slot_of[e]], 1),) + pk.row_shape, dtype=pk.data.dtype, device=dev))
                self._what.append(("edge", e, "edge_attr"))
        n_items = len(items)
        self._items = (_lib.CollateItem * n_items)(*items)
        self._slot_ptr = (C.c_void_p * len(slots))(*[p.ctypes.data for p in host_ptrs])
        h = C.c_void_p()
        _lib.check(lib.hmp_collator_create(len(slots), self._slot_ptr, store.n_graphs, n_items, self._items, C.byref(h)))
        self._h = h
        self._dst = (C.c_void_p * n_items)(*[b.data_ptr() for b in self._bufs])
        self._caps = (C.c_int64 * n_items)(*[self.cap[it.slot] for it in items])
        self._totals = (C.c_int64 * len(slots))()
        # Batch.ptr of every slot, written by the collation kernel (the executor's graph-local launch reads the node types' rows)
        self._off_stride = self.B + 1
        self._offsets = torch.zeros(len(slots) * self._off_stride, dtype=torch.int64, device=dev)
        self._slots, self._slot_of = slots, slot_of
        # ---- the executor's descriptor, filled once; per batch only the counts change
        from .engine import _BatchHolder

        nat = net.native() if hasattr(net, "native") else net
        self.nat = nat
        hd = _BatchHolder()
        buf_of = {w: b for w, b in zip(self._what, self._bufs)}
        self._node_slot, self._edge_slot = [], []
        for i, t in enumerate(nat.node_types):
            if t not in slot_of:
                raise _lib.HydraMPError(f"the store holds no node type '{t}'")
            self._node_slot.append(slot_of[t])
            if nat.in_dims.get(t, 0) > 0:
                x = buf_of[("node", t, "x")]
                if x.dtype != torch.float32 or x.size(1) != nat.in_dims[t]:
                    raise _lib.HydraMPError(f"store features of '{t}' are {tuple(x.shape[1:])} {x.dtype}, model expects {nat.in_dims[t]} float32")
                hd.c.d_x[i] = x.data_ptr()
                hd.c.ldx[i] = x.size(1)
        for i, e in enumerate(nat.edge_types):
            if e not in slot_of:
                raise _lib.HydraMPError(f"the store holds no edge type {e}")
            self._edge_slot.append(slot_of[e])
            hd.c.d_edge_index[i] = buf_of[("edge", e, "edge_index")].data_ptr()
            if nat.edge_dims.get(e, 0):
                hd.c.d_edge_attr[i] = buf_of[("edge", e, "edge_attr")].data_ptr()
        lab = buf_of[("node", label_type, label_key)]
        if lab.dtype != torch.int64:
            raise _lib.HydraMPError("labels in the store must be int64")
        hd.c.d_labels = lab.data_ptr()
        hd.keep = list(self._bufs) + [self._offsets]
        for i, t in enumerate(nat.node_types):
            hd.c.d_node_ptr[i] = self._offsets.data_ptr() + 8 * slot_of[t] * self._off_stride
        for i, e in enumerate(nat.edge_types):  # the collator's edge offsets: edges arrive graph by graph
            hd.c.d_edge_ptr[i] = self._offsets.data_ptr() + 8 * slot_of[e] * self._off_stride
        hd.c.max_graph_nodes = int(max(int(np.diff(host_ptrs[slot_of[t]]).max()) for t in store.node_types))
        hd.n_nodes = [0] * len(nat.node_types)
        hd.n_edges = [0] * len(nat.edge_types)
        out_type = nat.pool_edge_type[2] if nat.pool_edge_type is not None else nat.readout
        if out_type != label_type:
            raise _lib.HydraMPError(f"labels of '{label_type}' for a model that reads out '{out_type}'")
        self._out_slot = slot_of[out_type]
        self.holder = hd
        self.label_buf = lab
        # workspace sized once for the largest batch this stream can produce
        cap_h = _BatchHolder()
        cap_h.n_nodes = [self.cap[s] for s in self._node_slot]
        cap_h.n_edges = [self.cap[s] for s in self._edge_slot]
        flat = nat.flat_params()
        with torch.cuda.device(flat.device):
            nat._ensure_workspace(cap_h, flat.device)

    def next(self, ids) -> "object":
        """collate graphs ``ids`` (len == batch_size or fewer) into the stream's buffers; returns the descriptor for TrainStep.run"""
        sel = np.ascontiguousarray(ids, dtype=np.int32)
        B = int(sel.size)
        if B > self.B:
            raise _lib.HydraMPError(f"{B} graphs for a stream of batch size {self.B}")
        _lib.check(self.store.lib.hmp_collator_run(self._h, sel.ctypes.data, B, self._dst, self._caps, self._totals,
                                                   self._offsets.data_ptr(), self._off_stride, _lib.stream_ptr()))
        hd, tot = self.holder, self._totals
        for i, s in enumerate(self._node_slot):
            v = tot[s]
            hd.c.n_nodes[i] = v
            hd.n_nodes[i] = v
        for i, s in enumerate(self._edge_slot):
            v = tot[s]
            hd.c.n_edges[i] = v
            hd.n_edges[i] = v
        hd.c.n_out = tot[self._out_slot]
        hd.c.n_graphs = B
        self.num_graphs = B
        return hd

    def data(self) -> HeteroData:
        """the current batch as a HeteroData of VIEWS into the stream's buffers (valid until the next :meth:`next`)"""
        out = HeteroData()
        out.num_graphs = self.num_graphs
        for (kind, key, name), buf in zip(self._what, self._bufs):
            n = int(self._totals[self._slot_of[key]])
            if name == "edge_index":
                out[key].edge_index = buf[: 2 * n].view(2, n)
            else:
                setattr(out[key], name, buf[:n])
        return out

    def close(self):
        if self._h is not None:
            self.store.lib.hmp_collator_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _graphstore_stream(self, net, batch_size: int, label_type: str, label_key: str = "y") -> BatchStream:
    return BatchStream(self, net, batch_size, label_type, label_key)


GraphStore.stream = _graphstore_stream
