"""Receptive-field training step for small batches on large graphs.

The reference recomputes the embeddings of ALL nodes for every batch (``trainer.py:219``), also
at its default batch of 1 024 pairs, whose loss reads at most 7 168 rows of them.  For a 2-layer
model those rows depend only on their in-neighbours' hidden rows, and those only on THEIR
in-neighbours' input features.  This module computes exactly that closure with the same kernels:

* layer 1 on the row subset N1 = T u in(T) ("rectangular": the gather reads the global feature
  table, the outputs are compact),
* layer 2 on the rows T only, over the compact CSR of their in-edges (sources = positions in N1),
* the loss on pair indices remapped into T; autograd runs the backward on the compact tensors.

Loss and parameter gradients equal the full-graph step's (same terms, other summation order).
Opt-in (``TrainStep(..., receptive_field=True)``); it pays when 7*B rows and their two hops are a
small part of the graph.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib
from .graph import GraphCSR, build_csr, default_cache
from .ops import RowSubsetMean, SAGELayerFunction, aggregate_mean, segment_gather


def _in_edges(graph: GraphCSR, rows: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """CSR of the in-edges of ``rows`` (int64, any order): (rowptr int32 [n+1], source ids int32 [E],
    degrees int64 [n]); sources keep the global numbering and the CSR order of the full graph."""
    start = graph.rowptr[rows].to(torch.int64)
    deg = graph.rowptr[rows + 1].to(torch.int64) - start
    rp = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(deg, 0, out=rp[1:])
    total = int(rp[-1].item())
    if total == 0:
        return rp.to(torch.int32), torch.zeros(0, dtype=torch.int32, device=rows.device), deg
    pos = torch.repeat_interleave(start - rp[:-1], deg, output_size=total) + \
        torch.arange(total, device=rows.device)
    return rp.to(torch.int32), graph.col[pos], deg


def _rect_graph(n_rows: int, inv_deg: torch.Tensor) -> GraphCSR:
    """Stand-in graph object for a layer whose aggregation is handed in precomputed and whose
    input needs no gradient: only ``num_nodes`` and ``inv_deg`` are looked at."""
    e = torch.zeros(0, dtype=torch.int32, device=inv_deg.device)
    return GraphCSR(n_rows, 0, e, e, e, e, inv_deg, e, e, e, e)


def receptive_forward(model, x: torch.Tensor, graph: GraphCSR, targets: torch.Tensor,
                      contexts: torch.Tensor, negatives: torch.Tensor):
    """-> (embeddings of the rows T the batch reads [nT, D], targets', contexts', negatives'
    remapped into T).  2-layer models only."""
    if len(model.convs) != 2:
        raise ValueError("the receptive-field step covers 2-layer models")
    B = targets.numel()
    ids = torch.cat([targets, contexts, negatives.reshape(-1)])
    if B and (int(ids.min()) < 0 or int(ids.max()) >= graph.num_nodes):   # the ids index the CSR here
        raise IndexError("pair/negative ids outside the embedding table")
    T, inv = torch.unique(ids, return_inverse=True)            # rows the loss reads
    nT = T.numel()
    rp2, src2, deg2 = _in_edges(graph, T)
    N1, inv1 = torch.unique(torch.cat([T, src2.to(torch.int64)]), return_inverse=True)
    posT, col2 = inv1[:nT], inv1[nT:]
    n1 = N1.numel()
    # ---- layer 1 on the rows N1, sources read from the global feature table
    conv1, conv2 = model.convs[0], model.convs[1]
    w_l, w_r = conv1.lin_l.weight, conv1.lin_r.weight
    from .graphsage import vector_width
    xg = x
    width = vector_width(x.size(1))
    if width != x.size(1):
        xg = default_cache().padded(x, width)
        pad = width - x.size(1)
        w_l, w_r = torch.nn.functional.pad(w_l, (0, pad)), torch.nn.functional.pad(w_r, (0, pad))
    inv_deg1 = graph.inv_deg[N1]
    if conv1.cache_aggregation and not x.requires_grad:
        # the input features' neighbour mean is cached for the whole graph: pick the rows
        agg1 = default_cache().aggregated(x, xg, graph, aggregate_mean)[N1]
    else:
        rp1, col1, deg1 = _in_edges(graph, N1)
        thr = _lib.load().gsage_long_row_threshold()
        long1 = torch.nonzero(deg1 > thr).to(torch.int32).view(-1)
        n_long1 = torch.tensor([long1.numel()], dtype=torch.int32, device=x.device)
        agg1 = segment_gather(rp1, col1, xg, row_scale=inv_deg1,
                              long_rows=long1 if long1.numel() else None,
                              n_long=n_long1 if long1.numel() else None)
    x1 = xg[N1]
    p = float(model.dropout_rate)
    if model.training and p > 0.0:
        epi, seed = _lib.EPI_RELU_DROPOUT, model._dropout_seed()
    else:
        epi, seed, p = _lib.EPI_RELU, 0, 0.0
    h1 = SAGELayerFunction.apply(x1, w_l, conv1.lin_l.bias, w_r, _rect_graph(n1, inv_deg1), epi, p,
                                 seed, None, False, agg1)
    # ---- layer 2 on the rows T only: compact CSR (targets = positions in T, sources = positions
    #      in N1), mean over it with its own backward, projection with the aggregate handed in
    r_local = torch.repeat_interleave(torch.arange(nT, device=x.device), deg2,
                                      output_size=int(col2.numel()))
    g2 = build_csr(torch.stack([col2, r_local]), n1, check=False)
    agg2 = RowSubsetMean.apply(h1, g2, nT)
    root2 = h1[posT]
    epi2 = _lib.EPI_L2NORM if model.normalize_output else _lib.EPI_NONE
    emb = SAGELayerFunction.apply(root2, conv2.lin_l.weight, conv2.lin_l.bias, conv2.lin_r.weight,
                                  _rect_graph(nT, g2.inv_deg[:nT]), epi2, 0.0, 0, None, False, agg2)
    return emb, inv[:B], inv[B:2 * B], inv[2 * B:].view_as(negatives)
