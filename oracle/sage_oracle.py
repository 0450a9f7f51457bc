"""CPU oracle for the GraphSAGE training hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``graphsage_amd``) never does and
fails loudly when its HIP library is missing.

This is a restatement, in plain CPU torch ops, of what the reference executes on
its hot path (citations relative to the reference tree):

* ``src/model/graphsage.py:116-144`` -- layer order, ReLU/dropout placement,
  ``F.normalize`` on the last layer, ``num_layers`` handling ``:100-111``.
* third-party ``torch_geometric.nn.SAGEConv(in, out, aggr='mean')`` (PyG >= 2.3,
  unpinned: ``README.md:270``, ``setup.py:16``; constructed at
  ``graphsage.py:103,107,111``; called ``:133,138``).  PyG is not installable
  here, so its published algorithm for a ``Tensor`` edge_index is restated:
  ``msg = x.index_select(0, ei[0])`` ->
  ``zeros(N,F).scatter_add_(0, ei[1] expanded, msg)`` -> divide by
  ``zeros(N).scatter_add_(0, ei[1], ones).clamp(min=1)`` ->
  ``F.linear(agg, W_l, b_l) + F.linear(x, W_r)``.  The aggregation direction is
  stated a second time in-tree at ``src/model/layers.py:191-195``.
* ``src/model/loss.py:97-136`` and ``:161-193`` -- skip-gram loss and details.
* ``src/training/trainer.py:213-240`` -- the step body (zero_grad, forward, loss,
  backward, clip_grad_norm_(1.0), Adam(lr=1e-3)).

Parity pinning: the loss restatement is pinned by golden vectors generated from
the reference's own ``loss.py`` (``tests/golden/make_loss_golden.py``).  The
SAGEConv restatement is **parity unpinned** by the reference (its tests hold no
numeric values for this path, and PyG cannot be imported here); it is pinned by
the hand-computed known-answer tests in ``oracle/kat.py`` instead.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- CSR
def csr_build(edge_index: torch.Tensor, num_nodes: int):
    """Reference CSR for SURVEY row A10 (no reference function; degree definition
    is ``torch.bincount`` as at ``src/utils/graph_utils.py:38-39``).

    Returns numpy int32 ``rowptr[N+1], col[E]`` keyed on the destination
    (``edge_index[1]``), the transpose ``rowptr_t, col_t`` keyed on the source
    (``edge_index[0]``), both with edges of one row kept in ascending edge order
    (stable), and float32 ``inv_deg[N] = 1 / max(in_degree, 1)``.
    """
    ei = edge_index.detach().cpu().numpy().astype(np.int64)
    src, dst = ei[0], ei[1]
    n = int(num_nodes)

    def one(key, val):
        deg = np.bincount(key, minlength=n).astype(np.int64)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(deg, out=rowptr[1:])
        order = np.argsort(key, kind="stable")
        return rowptr.astype(np.int32), val[order].astype(np.int32), deg

    rowptr, col, deg_in = one(dst, src)
    rowptr_t, col_t, _ = one(src, dst)
    inv_deg = (1.0 / np.maximum(deg_in, 1).astype(np.float32)).astype(np.float32)
    return rowptr, col, rowptr_t, col_t, inv_deg


# ---------------------------------------------------------------------- SAGEConv
def mean_aggregate(x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """PyG ``scatter(x[ei[0]], ei[1], reduce='mean')`` (cf. layers.py:191-195)."""
    n, f = x.shape
    src, dst = edge_index[0], edge_index[1]
    msg = x.index_select(0, src)
    out = torch.zeros(n, f, dtype=x.dtype).scatter_add_(
        0, dst.view(-1, 1).expand(-1, f), msg)
    cnt = torch.zeros(n, dtype=x.dtype).scatter_add_(
        0, dst, torch.ones(dst.numel(), dtype=x.dtype)).clamp(min=1)
    return out / cnt.view(-1, 1)


def sage_conv(x, edge_index, w_l, b_l, w_r):
    """``SAGEConv.forward``: ``lin_l(mean_j x_j) + lin_r(x_i)`` (row A3)."""
    agg = mean_aggregate(x, edge_index)
    return F.linear(agg, w_l, b_l) + F.linear(x, w_r)


class OracleSAGEConv(nn.Module):
    """Parameter container with PyG's names: lin_l (weight+bias), lin_r (weight)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x, edge_index):
        return sage_conv(x, edge_index, self.lin_l.weight, self.lin_l.bias,
                         self.lin_r.weight)


class OracleGraphSAGE(nn.Module):
    """Restatement of ``GraphSAGE`` (graphsage.py:68-157)."""

    def __init__(self, in_channels=7, hidden_channels=64, out_channels=64,
                 num_layers=2, dropout=0.3, aggregator="mean",
                 normalize_output=True):
        super().__init__()
        if aggregator != "mean":
            raise ValueError("oracle restates aggr='mean' only")
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.out_channels, self.num_layers = out_channels, num_layers
        self.dropout_rate, self.normalize_output = dropout, normalize_output
        self.convs = nn.ModuleList()
        self.convs.append(OracleSAGEConv(in_channels, hidden_channels))  # :103
        for _ in range(num_layers - 2):                                    # :106
            self.convs.append(OracleSAGEConv(hidden_channels, hidden_channels))
        if num_layers > 1:                                                 # :110
            self.convs.append(OracleSAGEConv(hidden_channels, out_channels))
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, edge_index,
                keep_masks: Optional[Sequence[torch.Tensor]] = None):
        """``keep_masks`` (one {0,1} float tensor per hidden layer) replaces the
        Bernoulli draw of ``nn.Dropout`` so a kernel can be fed the same mask."""
        for i, conv in enumerate(self.convs[:-1]):                         # :132
            x = conv(x, edge_index)
            x = F.relu(x)
            if keep_masks is not None and self.training and self.dropout_rate > 0:
                x = x * keep_masks[i] / (1.0 - self.dropout_rate)
            else:
                x = self.dropout(x)
        x = self.convs[-1](x, edge_index)                                   # :138
        if self.normalize_output:
            x = F.normalize(x, p=2, dim=1)                                  # :142
        return x

    def count_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


# ------------------------------------------------------------------------- loss
def skipgram_loss(embeddings, targets, contexts, negatives, scale_factor=5.0):
    """``SkipGramLoss.forward`` (loss.py:97-136)."""
    t = embeddings[targets]
    c = embeddings[contexts]
    n = embeddings[negatives]
    pos = (t * c).sum(dim=1)
    neg = torch.bmm(n, t.unsqueeze(2)).squeeze(2)
    pos = pos * scale_factor
    neg = neg * scale_factor
    return -F.logsigmoid(pos).mean() + (-F.logsigmoid(-neg).mean())


def skipgram_loss_with_details(embeddings, targets, contexts, negatives,
                               scale_factor=5.0) -> Tuple[torch.Tensor, Dict[str, float]]:
    """``SkipGramLoss.forward_with_details`` (loss.py:161-195)."""
    t = embeddings[targets]
    c = embeddings[contexts]
    n = embeddings[negatives]
    pos_raw = (t * c).sum(dim=1)
    neg_raw = torch.bmm(n, t.unsqueeze(2)).squeeze(2)
    pos = pos_raw * scale_factor
    neg = neg_raw * scale_factor
    pos_loss = -F.logsigmoid(pos).mean()
    neg_loss = -F.logsigmoid(-neg).mean()
    total = pos_loss + neg_loss
    details = {
        "total_loss": total.item(),
        "pos_loss": pos_loss.item(),
        "neg_loss": neg_loss.item(),
        "pos_score_mean": pos_raw.mean().item(),
        "neg_score_mean": neg_raw.mean().item(),
        "pos_prob": torch.sigmoid(pos).mean().item(),
        "neg_prob": torch.sigmoid(neg).mean().item(),
    }
    return total, details


# ------------------------------------------------------------------ trainer step
def train_step(model: nn.Module, optimizer: torch.optim.Optimizer, x, edge_index,
               targets, contexts, negatives, scale_factor=5.0, gradient_clip=1.0,
               keep_masks=None):
    """One pass of the loop body at trainer.py:213-240 on CPU tensors."""
    optimizer.zero_grad()
    if keep_masks is not None:
        emb = model(x, edge_index, keep_masks=keep_masks)
    else:
        emb = model(x, edge_index)
    loss, details = skipgram_loss_with_details(emb, targets, contexts, negatives,
                                               scale_factor)
    loss.backward()
    if gradient_clip > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), gradient_clip)
    optimizer.step()
    return loss.detach(), details


def make_adam(model: nn.Module, lr=1e-3, weight_decay=0.0):
    """Optimizer as at trainer.py:149-153."""
    return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)


# ------------------------------------------------ feeder: alias table (next row)
def alias_table(edge_index: torch.Tensor, num_nodes: int, exponent: float = 0.75):
    """Restatement of ``DegreeBiasedNegativeSampler.__init__/_setup_alias_table``
    (src/walks/negative_sampler.py:64-145): out-degree on ``edge_index[0]``,
    +1e-8, ``deg**exponent`` normalised, Vose alias table with LIFO work lists.
    Returns float32 ``probs[N]``, float32 ``prob_table[N]``, int64 ``alias[N]``.
    """
    n = int(num_nodes)
    if edge_index.numel() == 0:
        degree = torch.ones(n, dtype=torch.float32)
    else:
        degree = torch.bincount(edge_index[0].cpu(), minlength=n).to(torch.float32)
        degree = degree + 1e-8
    w = degree ** exponent
    probs = w / w.sum()
    p = probs.numpy() * n            # float32 array scaled by python int
    alias = np.zeros(n, dtype=np.int64)
    prob_table = np.zeros(n, dtype=np.float32)
    smaller = [i for i in range(n) if p[i] < 1.0]
    larger = [i for i in range(n) if not (p[i] < 1.0)]
    while smaller and larger:
        s = smaller.pop()
        l = larger.pop()
        prob_table[s] = p[s]
        alias[s] = l
        p[l] = p[l] + p[s] - 1.0
        (smaller if p[l] < 1.0 else larger).append(l)
    for l in larger:
        prob_table[l] = 1.0
    for s in smaller:
        prob_table[s] = 1.0
    return probs, torch.from_numpy(prob_table), torch.from_numpy(alias)


# ------------------------------------------------ feeder: pair windows (next row, 8f-4)
def extract_pairs(walks, context_window: int):
    """Restatement of ``CooccurrencePairSampler.extract_pairs`` (src/walks/pair_sampler.py:56-86)."""
    pairs = []
    for walk in walks:
        n = len(walk)
        for i, target in enumerate(walk):
            for j in range(max(0, i - context_window), min(n, i + context_window + 1)):
                if i != j:
                    pairs.append((target, walk[j]))
    return pairs
