"""Graph structure cache: both CSR orientations of an ``edge_index`` on the device.

The reference rebuilds the scatter bookkeeping implicitly inside PyG on every
forward (``src/model/graphsage.py:133,138``).  Here it is built once per graph by the
K1 kernels (``gsage_csr_build``) and reused by every layer, forward and backward.
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib


@dataclass
class GraphCSR:
    num_nodes: int
    num_edges: int
    rowptr: torch.Tensor      # int32 [N+1]  in-edges grouped by target
    col: torch.Tensor         # int32 [E]    sources, stable edge order inside a row
    rowptr_t: torch.Tensor    # int32 [N+1]  out-edges grouped by source
    col_t: torch.Tensor       # int32 [E]    targets
    inv_deg: torch.Tensor     # fp32  [N]    1 / max(in_degree, 1)
    long_rows: torch.Tensor   # int32 [cap]  hub rows of (rowptr, col)
    long_rows_t: torch.Tensor  # int32 [cap]  hub rows of (rowptr_t, col_t)
    n_long: torch.Tensor      # int32 [2]
    status: torch.Tensor      # int32 [1]    device flag word

    def hidden1_groups(self) -> torch.Tensor:
        """The group index of gsage_segment_gather_hidden1 (rows' neighbours in groups of 32, per
        chunk of 32 rows), built on first use and kept with the CSR it describes."""
        g = getattr(self, "_h1_groups", None)
        if g is None:
            lib = _lib.load()
            with torch.cuda.device(self.rowptr.device):
                nbytes = lib.gsage_hidden1_groups_bytes(self.num_nodes, self.num_edges)
                g = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.rowptr.device)
                rc = lib.gsage_hidden1_groups_build(self.rowptr.data_ptr(), self.num_nodes, self.num_edges,
                                                    g.data_ptr(), nbytes, _lib.stream_handle())
            _lib.check(rc, "gsage_hidden1_groups_build")
            object.__setattr__(self, "_h1_groups", g)
        return g

    def in_degree(self) -> torch.Tensor:
        return (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)

    def out_degree(self) -> torch.Tensor:
        return (self.rowptr_t[1:] - self.rowptr_t[:-1]).to(torch.int64)


def build_csr(edge_index: torch.Tensor, num_nodes: int, check: bool = True) -> GraphCSR:
    """int64 ``[2,E]`` (row 0 = source, row 1 = target) -> :class:`GraphCSR`.

    ``check=True`` reads the device flag word once (one host sync per graph) and raises
    ``IndexError`` if any index lies outside ``[0, num_nodes)``.
    """
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
    if edge_index.dtype != torch.int64:
        raise ValueError(f"edge_index must be int64, got {edge_index.dtype}")
    if not edge_index.is_cuda:
        raise ValueError("graphsage_amd runs on the GPU only: edge_index must be a CUDA/HIP tensor")
    lib = _lib.load()
    ei = edge_index.contiguous()
    dev = ei.device
    E, N = int(ei.size(1)), int(num_nodes)
    i32 = dict(dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rowptr = torch.empty(N + 1, **i32)
        rowptr_t = torch.empty(N + 1, **i32)
        col = torch.empty(max(E, 1), **i32)
        col_t = torch.empty(max(E, 1), **i32)
        inv_deg = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        cap = E // max(lib.gsage_long_row_threshold(), 1) + 1
        long_rows = torch.empty(cap, **i32)
        long_rows_t = torch.empty(cap, **i32)
        n_long = torch.zeros(2, **i32)
        status = torch.zeros(1, **i32)
        ws_bytes = lib.gsage_csr_workspace_bytes(E, N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rc = lib.gsage_csr_build(ei.data_ptr(), E, N, rowptr.data_ptr(), col.data_ptr(),
                                 rowptr_t.data_ptr(), col_t.data_ptr(), inv_deg.data_ptr(),
                                 long_rows.data_ptr(), long_rows_t.data_ptr(), cap,
                                 n_long.data_ptr(), status.data_ptr(), ws.data_ptr(), ws_bytes,
                                 _lib.stream_handle())
        _lib.check(rc, "gsage_csr_build")
        if check and int(status.item()) & _lib.FLAG_INDEX_OUT_OF_RANGE:
            raise IndexError(f"edge_index holds node ids outside [0, {N})")
        del ws
    return GraphCSR(N, E, rowptr, col[:E], rowptr_t, col_t[:E], inv_deg[:N], long_rows,
                    long_rows_t, n_long, status)


class GraphCache:
    """Keeps the CSR of the graphs recently seen, keyed on the identity of the ``edge_index``
    tensor OBJECT (held weakly) and its in-place version counter.  A device pointer alone is
    not an identity: the caching allocator hands the same address to the next graph.

    Three separate stores, so that one kind of entry never evicts another: CSRs (``capacity``
    entries, least recently USED goes first), padded input features, cached input aggregates
    (``aux_capacity`` each).  Entries whose key tensor has died are swept on every insert."""

    def __init__(self, capacity: int = 4, aux_capacity: int = 4):
        self.capacity = capacity
        self.aux_capacity = aux_capacity
        self._graphs: "OrderedDict[int, Tuple]" = OrderedDict()
        self._pads: "OrderedDict[Tuple, Tuple]" = OrderedDict()
        self._aggs: "OrderedDict[Tuple, Tuple]" = OrderedDict()
        self.builds = 0                 # CSR builds so far (tests, diagnostics)

    @staticmethod
    def _sweep(store) -> None:
        for dead in [k for k, e in store.items() if any(r() is None for r in e[0])]:
            del store[dead]

    @staticmethod
    def _insert(store, key, entry, capacity: int) -> None:
        GraphCache._sweep(store)
        while len(store) >= capacity:
            store.popitem(last=False)
        store[key] = entry

    def get(self, edge_index: torch.Tensor, num_nodes: int) -> GraphCSR:
        k = id(edge_index)
        ent = self._graphs.get(k)
        if ent is not None:
            (ref,), version, n, g = ent
            if ref() is edge_index and version == edge_index._version and n == int(num_nodes):
                self._graphs.move_to_end(k)
                return g
            del self._graphs[k]
        g = build_csr(edge_index, num_nodes)
        self.builds += 1
        self._insert(self._graphs, k, ((weakref.ref(edge_index),), edge_index._version,
                                       int(num_nodes), g), self.capacity)
        return g

    def padded(self, x: torch.Tensor, width: int, keep: bool = True) -> torch.Tensor:
        """``x`` zero-padded to ``width`` columns.  With ``keep`` the result is cached per tensor
        object + version when x needs no gradient: for the INPUT features, which are constant
        across training steps.  Callers padding a temporary (a hidden activation in eval mode)
        pass ``keep=False``: such a tensor never comes back, caching it would only hold memory."""
        pad = width - x.size(1)
        if x.requires_grad or not keep:
            return torch.nn.functional.pad(x, (0, pad))
        k = (id(x), width)
        ent = self._pads.get(k)
        if ent is not None:
            (ref,), version, xp = ent
            if ref() is x and version == x._version:
                self._pads.move_to_end(k)
                return xp
            del self._pads[k]
        xp = torch.nn.functional.pad(x, (0, pad)).contiguous()
        self._insert(self._pads, k, ((weakref.ref(x),), x._version, xp), self.aux_capacity)
        return xp

    def aggregated(self, x_key: torch.Tensor, x: torch.Tensor, graph: "GraphCSR", compute):
        """``compute(graph, x)`` (the mean aggregation of the INPUT features), cached per
        (feature tensor object + version, graph): node features and graph are constant across
        training steps, so an opt-in caller (``GraphSAGE(cache_input_aggregation=True)``) can skip
        the first layer's gather after the first step.  ``x_key`` is the caller's tensor, ``x``
        the (possibly padded) one that is aggregated."""
        k = (id(x_key), id(graph))
        ent = self._aggs.get(k)
        if ent is not None:
            (ref, gref), version, agg = ent
            if ref() is x_key and version == x_key._version and gref() is graph:
                self._aggs.move_to_end(k)
                return agg
            del self._aggs[k]
        agg = compute(graph, x)
        self._insert(self._aggs, k, ((weakref.ref(x_key), weakref.ref(graph)), x_key._version, agg),
                     self.aux_capacity)
        return agg

    def held_for(self, features: torch.Tensor, edge_index: torch.Tensor) -> list:
        """Strong references to everything cached for this (features, edge_index) pair: the CSR,
        the padded features, the cached aggregate.  A captured HIP graph has their device
        addresses baked in, so whoever owns that graph must keep these alive itself rather than
        rely on this cache not evicting them (TrainStep.capture_step does)."""
        out = []
        g = None
        ent = self._graphs.get(id(edge_index))
        if ent is not None and ent[0][0]() is edge_index:
            g = ent[3]
            out.append(g)
        for (refs, _, val) in self._pads.values():
            if refs[0]() is features:
                out.append(val)
        for (refs, _, val) in self._aggs.values():
            if refs[0]() is features and (g is None or refs[1]() is g):
                out.append(val)
        return out

    def clear(self) -> None:
        self._graphs.clear()
        self._pads.clear()
        self._aggs.clear()


_default_cache: Optional[GraphCache] = None


def default_cache() -> GraphCache:
    global _default_cache
    if _default_cache is None:
        _default_cache = GraphCache()
    return _default_cache
