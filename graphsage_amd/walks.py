"""Random walks and co-occurrence pairs on the device: counterparts of
``RandomWalkGenerator`` (``src/walks/generator.py``) and ``CooccurrencePairSampler``
(``src/walks/pair_sampler.py``).

The reference builds a Python dict adjacency with a loop over every edge, walks with
``random.choice`` and extracts pairs with a triple Python loop; a 10 M-node graph at the default
80x10 walks would give 1.5e11 pairs, which cannot be materialised anywhere.  Here walks live in
one int32 ``[W, L]`` tensor, and pairs are addressed by their number in the reference's own
enumeration order, so they can be extracted in bulk (small graphs, exact reference order) or
sampled uniformly per batch (large graphs) by the same kernel.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _lib
from .graph import GraphCSR, default_cache


class RandomWalkGenerator:
    def __init__(self, edge_index: torch.Tensor, num_nodes: int, walk_length: int = 80,
                 walks_per_node: int = 10, seed: Optional[int] = 42):
        self.num_nodes, self.walk_length, self.walks_per_node = num_nodes, walk_length, walks_per_node
        self.seed = 0 if seed is None else int(seed)
        self.graph: GraphCSR = default_cache().get(edge_index, num_nodes)
        self._edge_index = edge_index         # keeps the cache entry's key alive

    def generate_all_walks(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(walks int32 [N*walks_per_node, walk_length] padded with -1, lengths int32 [W])."""
        lib = _lib.load()
        g = self.graph
        dev = g.rowptr_t.device
        W = self.num_nodes * self.walks_per_node
        with torch.cuda.device(dev):
            walks = torch.empty(W, self.walk_length, dtype=torch.int32, device=dev)
            lengths = torch.empty(W, dtype=torch.int32, device=dev)
            rc = lib.gsage_random_walks(g.rowptr_t.data_ptr(), g.col_t.data_ptr(), self.num_nodes,
                                        self.walk_length, self.walks_per_node,
                                        self.seed & 0xFFFFFFFFFFFFFFFF, walks.data_ptr(),
                                        lengths.data_ptr(), _lib.stream_handle())
        _lib.check(rc, "gsage_random_walks")
        return walks, lengths

    @staticmethod
    def to_lists(walks: torch.Tensor, lengths: torch.Tensor) -> List[List[int]]:
        """The reference's list-of-lists form (walks of a single node dropped, generator.py:165)."""
        w, n = walks.cpu().tolist(), lengths.cpu().tolist()
        return [row[:k] for row, k in zip(w, n) if k > 1]


class CooccurrencePairSampler:
    def __init__(self, context_window: int = 10):
        self.context_window = context_window

    def index(self, walks: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """int64 [W+1] offsets: offsets[w] = number of pairs of walks 0..w-1; offsets[-1] = total."""
        lib = _lib.load()
        W = lengths.numel()
        dev = lengths.device
        with torch.cuda.device(dev):
            counts = torch.empty(W, dtype=torch.int64, device=dev)
            rc = lib.gsage_walk_pair_counts(lengths.data_ptr(), W, self.context_window,
                                            counts.data_ptr(), _lib.stream_handle())
        _lib.check(rc, "gsage_walk_pair_counts")
        offsets = torch.zeros(W + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=offsets[1:])
        return offsets

    def _gather(self, walks, lengths, offsets, pair_ids, first_id, count):
        lib = _lib.load()
        dev = walks.device
        with torch.cuda.device(dev):
            t = torch.empty(count, dtype=torch.int64, device=dev)
            c = torch.empty(count, dtype=torch.int64, device=dev)
            rc = lib.gsage_walk_pairs_gather(walks.data_ptr(), lengths.data_ptr(), walks.size(1),
                                             offsets.data_ptr(), lengths.numel(),
                                             _lib.ptr(pair_ids), first_id, count,
                                             self.context_window, t.data_ptr(), c.data_ptr(),
                                             _lib.stream_handle())
        _lib.check(rc, "gsage_walk_pairs_gather")
        return t, c

    def extract_pairs_tensor(self, walks: torch.Tensor, lengths: torch.Tensor,
                             max_pairs: int = 1 << 31) -> Tuple[torch.Tensor, torch.Tensor]:
        """All pairs, in the reference's order (``pair_sampler.py:88-120``)."""
        offsets = self.index(walks, lengths)
        total = int(offsets[-1].item())
        if total > max_pairs:
            raise MemoryError(f"{total} pairs: sample them per batch with sample_pairs() instead")
        if total == 0:
            z = torch.zeros(0, dtype=torch.int64, device=walks.device)
            return z, z.clone()
        return self._gather(walks, lengths, offsets, None, 0, total)

    def sample_pairs(self, walks: torch.Tensor, lengths: torch.Tensor, offsets: torch.Tensor,
                     batch_size: int, generator: Optional[torch.Generator] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
        """``batch_size`` pairs drawn uniformly (with replacement) from ALL pairs of the walks."""
        total = int(offsets[-1].item())
        ids = torch.randint(0, total, (batch_size,), device=walks.device, generator=generator)
        return self._gather(walks, lengths, offsets, ids, 0, batch_size)


def estimate_pair_count(num_walks: int, avg_walk_length: float, context_window: int) -> int:
    """``pair_sampler.py:275-307``: walks x length x 2 x window (an upper estimate)."""
    return int(num_walks * avg_walk_length * 2 * context_window)
