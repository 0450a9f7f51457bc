"""Synthetic inputs for tests and bench: power-law "memory graphs" in
the layout the reference's view extractor emits (``src/data/view_extractor.py:182-188``:
symmetric, de-duplicated, sorted by (src, dst), int64 ``[2, E]``), 7-column features shaped
like ``src/data/feature_extractor.py:124-165``, and co-occurrence pair batches with
``deg^0.75`` negatives (``src/walks/negative_sampler.py:84-90``).  Pure torch (plumbing, not
product); runs on whichever device it is given.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def powerlaw_graph(num_nodes: int, avg_degree: float = 20.0, gamma: float = 2.5,
                   seed: int = 42, device="cpu", max_degree: Optional[int] = None,
                   shuffle_ids: bool = True) -> torch.Tensor:
    """Chung-Lu graph: expected degree of node i ~ (i + i0)^(-1/(gamma-1)), capped at
    ``max_degree`` (default 4*sqrt(N*avg_degree)).  Returns edge_index int64 [2, E] with
    E ~ N*avg_degree directed edges (both directions present, no self loops, no duplicates)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = int(num_nodes)
    if max_degree is None:
        max_degree = int(4.0 * (n * avg_degree) ** 0.5)
    alpha = 1.0 / (gamma - 1.0)
    w = (torch.arange(n, dtype=torch.float64, device=dev) + 1.0).pow(-alpha)
    w = w * (n * avg_degree / w.sum())
    for _ in range(8):  # water-fill the cap so the mean degree is kept
        w = w.clamp(max=float(max_degree))
        w = w * (n * avg_degree / w.sum())
    w = w.clamp(max=float(max_degree))
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1].clone()
    m = int(n * avg_degree / 2)
    chunks = []
    step = 1 << 26
    for lo in range(0, m, step):
        k = min(step, m - lo)
        a = torch.searchsorted(cdf, torch.rand(k, dtype=torch.float64, device=dev, generator=g))
        b = torch.searchsorted(cdf, torch.rand(k, dtype=torch.float64, device=dev, generator=g))
        a.clamp_(max=n - 1)
        b.clamp_(max=n - 1)
        keep = a != b
        chunks.append(torch.stack([a[keep], b[keep]]))
    und = torch.cat(chunks, dim=1) if chunks else torch.zeros(2, 0, dtype=torch.int64, device=dev)
    del chunks
    if shuffle_ids:
        perm = torch.randperm(n, device=dev, generator=g)
        und = perm[und]
    key = torch.cat([und[0] * n + und[1], und[1] * n + und[0]])
    del und
    key = torch.unique(key)  # sorted => lexicographic (src, dst)
    return torch.stack([key // n, key % n]).contiguous()


def memory_features(num_nodes: int, seed: int = 42, device="cpu", dim: int = 7) -> torch.Tensor:
    """fp32 [N, dim]: first 4 columns ~ U[0,1], remaining columns Bernoulli(0.5)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1)
    x = torch.rand(num_nodes, dim, device=dev, generator=g)
    if dim > 4:
        x[:, 4:] = (x[:, 4:] < 0.5).to(torch.float32)
    return x.contiguous()


def cooccurrence_batch(edge_index: torch.Tensor, num_nodes: int, batch_size: int,
                       num_negatives: int = 5, seed: int = 1, exponent: float = 0.75
                       ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(targets [B], contexts [B], negatives [B,K]) int64 on edge_index's device.
    target = source of a random edge (never isolated); context = its 1-hop neighbour on that
    edge, or with probability 1/2 a neighbour of that neighbour (2-hop, walk-like locality);
    negatives ~ out_degree^0.75 by inverse-CDF sampling."""
    dev = edge_index.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    E = edge_index.size(1)
    src, dst = edge_index[0], edge_index[1]
    e1 = torch.randint(0, E, (batch_size,), device=dev, generator=g)
    targets = src[e1]
    hop1 = dst[e1]
    # a random out-edge of hop1: edges are sorted by src, so use searchsorted on src
    deg = torch.bincount(src, minlength=num_nodes)
    start = torch.cumsum(deg, 0) - deg
    r = torch.rand(batch_size, device=dev, generator=g)
    off = (r * deg[hop1].to(torch.float32)).to(torch.int64).clamp_(max=(deg[hop1] - 1).clamp(min=0))
    hop2 = dst[(start[hop1] + off).clamp_(max=E - 1)]
    two = torch.rand(batch_size, device=dev, generator=g) < 0.5
    contexts = torch.where(two & (deg[hop1] > 0), hop2, hop1)
    wts = (deg.to(torch.float64) + 1e-8).pow(exponent)
    cdf = torch.cumsum(wts, 0)
    cdf = cdf / cdf[-1].clone()
    u = torch.rand(batch_size * num_negatives, dtype=torch.float64, device=dev, generator=g)
    negatives = torch.searchsorted(cdf, u).clamp_(max=num_nodes - 1).view(batch_size, num_negatives)
    return targets.contiguous(), contexts.contiguous(), negatives.contiguous()
