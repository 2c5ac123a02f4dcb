"""Scene-graph containers that honour the PyG ``HeteroData`` / ``Data`` / ``Batch`` input contract.

The reference hands its models a ``torch_geometric.data.HeteroData`` (or a ``Batch`` of them)
and reads only a handful of accessors from it:

* ``data.x_dict`` / ``data.edge_index_dict`` / ``data.edge_attr_dict``
  (reference ``src/hydra_gnn/models/heterogeneous_network.py:99-106``),
* ``data["room_virtual"].num_nodes`` (``heterogeneous_neural_tree_network.py:184``),
* ``batch["rooms"].y`` / ``batch["room_virtual"].y`` (``base_training_job.py:210-212``),
* ``data.x`` / ``data.edge_index`` / ``data.room_mask`` / ``data.edge_attr`` for homogeneous
  graphs (``homogeneous_network.py:122-128``),
* ``.to(device)``.

torch_geometric is not a dependency of this engine, so these minimal containers provide the same
accessors.  The models in :mod:`hydra_gnn_amd.models` are duck-typed: a genuine PyG
``HeteroData``/``Batch`` works as well.

Layout of a collated batch follows SURVEY.md Appendix B.3 (PyG ``Batch.from_data_list``):
per node type the node stores are concatenated in graph order; per edge type the ``edge_index``
columns are concatenated with the source row offset by the cumulative node count of the *source*
type and the destination row by that of the *destination* type.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import torch

EdgeType = Tuple[str, str, str]

# reference src/hydra_gnn/mp3d_dataset.py:21-26 (order matters: HeteroConv iterates in this order)
EDGE_TYPES: List[EdgeType] = [
    ("objects", "objects_to_objects", "objects"),
    ("rooms", "rooms_to_rooms", "rooms"),
    ("objects", "objects_to_rooms", "rooms"),
    ("rooms", "rooms_to_objects", "objects"),
]

# reference src/hydra_gnn/neural_tree/construct.py:14-38 (names incl. the `or_ro_rr` spelling are
# part of the state_dict contract)
HTREE_NODE_TYPES = ["object", "room", "object-room", "room-room"]
HTREE_EDGE_TYPES: List[EdgeType] = [
    ("object", "o_to_or", "object-room"),
    ("object-room", "or_to_o", "object"),
    ("room", "r_to_or", "object-room"),
    ("object-room", "or_to_r", "room"),
    ("room", "r_to_rr", "room-room"),
    ("room-room", "rr_to_r", "room"),
    ("object-room", "or_ro_rr", "room-room"),
    ("room-room", "rr_ro_or", "object-room"),
    ("object-room", "or_to_or", "object-room"),
    ("room-room", "rr_to_rr", "room-room"),
]
HTREE_VIRTUAL_NODE_TYPES = ["object_virtual", "room_virtual"]
HTREE_INIT_EDGE_TYPES: List[EdgeType] = [
    ("object_virtual", "ov_to_or", "object-room"),
    ("room_virtual", "rv_to_or", "object-room"),
    ("room_virtual", "rv_to_rr", "room-room"),
]
HTREE_POOL_EDGE_TYPES: List[EdgeType] = [
    ("object", "o_to_ov", "object_virtual"),
    ("room", "r_to_rv", "room_virtual"),
]


class _Store:
    """Attribute bag for one node type or one edge type."""

    def __init__(self) -> None:
        object.__setattr__(self, "_d", {})
        object.__setattr__(self, "_v", 0)  # bumped on every attribute assignment (lets the engine cache a batch descriptor)

    def __getattr__(self, name):
        d = object.__getattribute__(self, "_d")
        if name in d:
            return d[name]
        if name == "num_nodes":
            if "x" in d:
                return int(d["x"].size(0))
            for v in d.values():
                if isinstance(v, torch.Tensor) and v.dim() >= 1:
                    return int(v.size(0))
            return 0
        if name == "num_edges":
            return int(d["edge_index"].size(1)) if "edge_index" in d else 0
        if name == "num_node_features":
            return int(d["x"].size(1)) if "x" in d else 0
        raise AttributeError(name)

    def __setattr__(self, name, value):
        d = object.__getattribute__(self, "_d")
        if name == "edge_index":
            # `ptr` / `ptr_version` describe the edge_index they were computed for (collate): a new edge_index drops them, so the
            # engine never vouches for a graph-sorted edge list that was replaced after collation
            d.pop("ptr", None)
            d.pop("ptr_version", None)
        d[name] = value
        object.__setattr__(self, "_v", object.__getattribute__(self, "_v") + 1)

    def __delattr__(self, name):
        d = object.__getattribute__(self, "_d")
        if name not in d:
            raise AttributeError(name)
        del d[name]
        object.__setattr__(self, "_v", object.__getattribute__(self, "_v") + 1)

    def __contains__(self, name):
        return name in object.__getattribute__(self, "_d")

    def keys(self):
        return object.__getattribute__(self, "_d").keys()

    def items(self):
        return object.__getattribute__(self, "_d").items()

    def _map(self, fn):
        out = _Store()
        for k, v in self.items():
            setattr(out, k, fn(v) if isinstance(v, torch.Tensor) else v)
        if "ptr_version" in out and "edge_index" in out:  # the copy starts its own version history
            stale = "edge_index" in self and int(self.edge_index._version) != int(self.ptr_version)
            if stale:
                delattr(out, "ptr")
                delattr(out, "ptr_version")
            else:
                out.ptr_version = int(out.edge_index._version)
        return out


class HeteroData:
    """Minimal stand-in for ``torch_geometric.data.HeteroData``."""

    def __init__(self) -> None:
        self._nodes: Dict[str, _Store] = {}
        self._edges: Dict[EdgeType, _Store] = {}
        self._plan_cache = None  # filled lazily by hydra_gnn_amd.engine (CSR/CSC plan)

    # -- store access -------------------------------------------------------------------------
    def __getitem__(self, key: Union[str, EdgeType]) -> _Store:
        if isinstance(key, tuple):
            key = tuple(key)
            if key not in self._edges:
                self._edges[key] = _Store()
            return self._edges[key]
        if key not in self._nodes:
            self._nodes[key] = _Store()
        return self._nodes[key]

    def _mutation_stamp(self) -> int:
        """changes whenever a store is added or an attribute of any store is (re)assigned"""
        v = len(self._nodes) + len(self._edges)
        for s in self._nodes.values():
            v += object.__getattribute__(s, "_v") << 8
        for s in self._edges.values():
            v += object.__getattribute__(s, "_v") << 8
        return v

    @property
    def node_types(self) -> List[str]:
        return list(self._nodes.keys())

    @property
    def edge_types(self) -> List[EdgeType]:
        return list(self._edges.keys())

    # the PyG accessors build a fresh dict on every access; the reference relies on that
    # (heterogeneous_neural_tree_network.py:155-159)
    @property
    def x_dict(self) -> Dict[str, torch.Tensor]:
        return {k: s.x for k, s in self._nodes.items() if "x" in s}

    @property
    def edge_index_dict(self) -> Dict[EdgeType, torch.Tensor]:
        return {k: s.edge_index for k, s in self._edges.items() if "edge_index" in s}

    @property
    def edge_attr_dict(self) -> Dict[EdgeType, torch.Tensor]:
        return {k: s.edge_attr for k, s in self._edges.items() if "edge_attr" in s}

    def to(self, device) -> "HeteroData":
        out = type(self)()
        out._nodes = {k: s._map(lambda t: t.to(device)) for k, s in self._nodes.items()}
        out._edges = {k: s._map(lambda t: t.to(device)) for k, s in self._edges.items()}
        for name in ("num_graphs", "max_graph_nodes"):
            if hasattr(self, name):
                setattr(out, name, getattr(self, name))
        return out


class Data:
    """Minimal stand-in for ``torch_geometric.data.Data`` (homogeneous graph)."""

    def __init__(self, **kw) -> None:
        self._plan_cache = None
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return int(self.x.size(0))

    def to(self, device) -> "Data":
        out = type(self)()
        for k, v in self.__dict__.items():
            if k == "_plan_cache":
                continue
            setattr(out, k, v.to(device) if isinstance(v, torch.Tensor) else v)
        return out


# ---------------------------------------------------------------------------------------------
# reference: Hydra_mp3d_data.fill_missing_edge_index, src/hydra_gnn/mp3d_dataset.py:220-255;
# known answers: tests/test_mp3d_dataset.py:143-180
# ---------------------------------------------------------------------------------------------
def fill_missing_edge_index(torch_data: HeteroData, edge_types: Sequence[EdgeType]) -> None:
    """Make every edge type of ``edge_types`` present.

    A missing inter-type relation whose reverse (names split on ``_`` and reversed, e.g.
    ``rooms_to_objects`` <-> ``objects_to_rooms``) exists becomes the row-flipped copy of that
    reverse; everything else missing becomes an empty ``[2, 0]`` int64 index.
    """
    present = torch_data.edge_index_dict
    for src, rel, dst in edge_types:
        if (src, rel, dst) in present:
            continue
        filled = torch.empty((2, 0), dtype=torch.int64)
        if src != dst:
            reverse = (dst, "_".join(reversed(rel.split("_"))), src)
            if reverse in present:
                filled = present[reverse].flip([0])
        torch_data[src, rel, dst].edge_index = filled


# reference: Hydra_mp3d_data.compute_relative_pos, src/hydra_gnn/mp3d_dataset.py:298-319
def compute_relative_pos(torch_data: HeteroData) -> None:
    """Drop the leading xyz columns of every ``x`` and set ``edge_attr = pos_dst[i] - pos_src[j]``."""
    if torch_data.edge_attr_dict:
        raise Warning("Cannot compute relative pos as edge_attr -- edge_attr is not empty.")
    for nt in torch_data.node_types:
        torch_data[nt].x = torch_data[nt].x[:, 3:]
    for (src, rel, dst), ei in torch_data.edge_index_dict.items():
        torch_data[src, rel, dst].edge_attr = torch_data[dst].pos[ei[1]] - torch_data[src].pos[ei[0]]


# ---------------------------------------------------------------------------------------------
# collate: PyG Batch.from_data_list semantics (reference caller
# src/hydra_gnn/base_training_job.py:164-178)
# ---------------------------------------------------------------------------------------------
def collate(graphs: Sequence[HeteroData]) -> HeteroData:
    assert len(graphs) > 0
    out = HeteroData()
    out.num_graphs = len(graphs)
    node_types = graphs[0].node_types
    offsets: Dict[str, List[int]] = {}
    max_nodes = 0
    for nt in node_types:
        counts = [g[nt].num_nodes for g in graphs]
        off = [0]
        for c in counts:
            off.append(off[-1] + c)
        offsets[nt] = off
        keys = [k for k in graphs[0][nt].keys() if isinstance(getattr(graphs[0][nt], k), torch.Tensor)]
        for k in keys:
            setattr(out[nt], k, torch.cat([getattr(g[nt], k) for g in graphs], dim=0))
        if "num_nodes" in graphs[0][nt] and "x" not in graphs[0][nt]:
            out[nt].num_nodes = off[-1]
        out[nt].batch = torch.repeat_interleave(
            torch.arange(len(graphs), dtype=torch.int64), torch.tensor(counts, dtype=torch.int64)
        )
        out[nt].ptr = torch.tensor(off, dtype=torch.int64)
        max_nodes = max(max_nodes, max(counts))
    for et in graphs[0].edge_types:
        src, _, dst = et
        parts = []
        for gi, g in enumerate(graphs):
            ei = g[et].edge_index
            shift = torch.tensor([[offsets[src][gi]], [offsets[dst][gi]]], dtype=ei.dtype)
            parts.append(ei + shift)
        out[et].edge_index = torch.cat(parts, dim=1)
        # edges of graph g are entries [ptr[g], ptr[g+1]) (what [PyG] keeps in Batch._slice_dict): lets the plan build of a
        # small batch read, per block of rows, only the edges of the graphs that own them (hmp_batch::d_edge_ptr)
        eoff = [0]
        for q in parts:
            eoff.append(eoff[-1] + q.size(1))
        out[et].ptr = torch.tensor(eoff, dtype=torch.int64)
        out[et].ptr_version = int(out[et].edge_index._version)  # an in-place edit of edge_index afterwards invalidates `ptr`
        if "edge_attr" in graphs[0][et]:
            out[et].edge_attr = torch.cat([g[et].edge_attr for g in graphs], dim=0)
    # host knowledge of the collation: lets the engine run the per-graph phases of a training step as one launch (hmp_batch)
    out.max_graph_nodes = int(max_nodes)
    return out


# reference: heterogeneous_data_to_homogeneous / heterogeneous_htree_to_homogeneous, src/hydra_gnn/mp3d_dataset.py:29-119
# ([PyG] HeteroData.to_homogeneous: node types concatenated in store order, every edge type's edge_index shifted by the
# offsets of its endpoint types and concatenated in edge-type order; node_type / edge_type id vectors).
def heterogeneous_data_to_homogeneous(torch_data: HeteroData):
    """-> (Data with x, edge_index, node_type, edge_type [, y, edge_attr], list of node type names).  Features are
    zero-padded to the widest node type (reference :36-51)."""
    node_types = [t for t in torch_data.node_types if "x" in torch_data[t]]
    width = max(torch_data[t].num_node_features for t in node_types)
    xs, ys, nts, off = [], [], [], {}
    n = 0
    has_y = all("y" in torch_data[t] for t in node_types)
    for i, t in enumerate(node_types):
        x = torch_data[t].x
        if x.size(0) == 0:
            x = torch.empty((0, width), dtype=x.dtype)
        xs.append(torch.nn.functional.pad(x, (0, width - x.size(1), 0, 0), mode="constant", value=0))
        nts.append(torch.full((x.size(0),), i, dtype=torch.int64))
        if has_y:
            ys.append(torch_data[t].y)
        off[t] = n
        n += x.size(0)
    eis, ets, eas = [], [], []
    edge_types = [e for e in torch_data.edge_types if "edge_index" in torch_data[e]]
    has_ea = len(edge_types) > 0 and all("edge_attr" in torch_data[e] for e in edge_types)
    for i, e in enumerate(edge_types):
        ei = torch_data[e].edge_index
        shift = torch.tensor([[off[e[0]]], [off[e[2]]]], dtype=ei.dtype)
        eis.append(ei + shift)
        ets.append(torch.full((ei.size(1),), i, dtype=torch.int64))
        if has_ea:
            eas.append(torch_data[e].edge_attr)
    out = Data(x=torch.cat(xs, 0), node_type=torch.cat(nts, 0),
               edge_index=torch.cat(eis, 1) if eis else torch.empty((2, 0), dtype=torch.int64),
               edge_type=torch.cat(ets, 0) if ets else torch.empty((0,), dtype=torch.int64))
    if has_y:
        out.y = torch.cat(ys, 0)
    if has_ea:
        out.edge_attr = torch.cat(eas, 0)
    return out, node_types


def heterogeneous_htree_to_homogeneous(torch_data: HeteroData) -> Data:
    """Augmented H-tree as one homogeneous graph (reference :73-119): ``edge_index`` keeps the H-tree edges,
    ``init_edge_index`` the virtual -> clique edges, ``pool_edge_index`` the leaf -> virtual edges; ``room_mask`` /
    ``object_mask`` mark the virtual nodes (the classification nodes)."""
    assert "object_virtual" in torch_data.x_dict and "room_virtual" in torch_data.x_dict
    if "y" in torch_data["room_virtual"]:
        y_dtype = torch_data["room_virtual"].y.dtype
        for t in torch_data.x_dict:
            if "y" not in torch_data[t]:
                torch_data[t].y = -torch.ones(torch_data[t].num_nodes, dtype=y_dtype)
    d, node_types = heterogeneous_data_to_homogeneous(torch_data)
    ov, rv = node_types.index("object_virtual"), node_types.index("room_virtual")
    src_t = d.node_type[d.edge_index[0]]
    dst_t = d.node_type[d.edge_index[1]]
    init_mask = (src_t == ov) | (src_t == rv)
    pool_mask = (dst_t == ov) | (dst_t == rv)
    d.init_edge_index = d.edge_index[:, init_mask]
    d.pool_edge_index = d.edge_index[:, pool_mask]
    keep = ~(init_mask | pool_mask)
    d.edge_index = d.edge_index[:, keep]
    d.edge_type = d.edge_type[keep]
    if hasattr(d, "edge_attr"):
        d.edge_attr = d.edge_attr[keep, :]
    d.room_mask = d.node_type == rv
    d.object_mask = d.node_type == ov
    return d


def collate_homogeneous(graphs: Sequence[Data]) -> Data:
    assert len(graphs) > 0
    out = Data()
    off = [0]
    for g in graphs:
        off.append(off[-1] + g.num_nodes)
    for k, v in graphs[0].__dict__.items():
        if k == "_plan_cache" or not isinstance(v, torch.Tensor):
            continue
        if "index" in k:  # PyG offsets every attribute whose name contains "index"
            out.__dict__[k] = torch.cat([getattr(g, k) + off[i] for i, g in enumerate(graphs)], dim=1)
        else:
            out.__dict__[k] = torch.cat([getattr(g, k) for g in graphs], dim=0)
    out.num_graphs = len(graphs)
    return out
