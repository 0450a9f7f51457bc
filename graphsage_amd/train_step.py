"""Minimal training-step harness: the loop body of the reference trainer
(``src/training/trainer.py:213-246``) around the MI355X hot path, plus the pair-sharded
data-parallel exchange the reference does not have.

Per step: zero_grad -> full-graph forward -> skip-gram loss on this rank's pair shard ->
backward -> [one all-reduce of the flat gradient buffer over RCCL] -> clip_grad_norm_ ->
Adam.  Graph, features and weights are replicated on every GPU; only pairs are sharded.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from .loss import SkipGramLoss, stats_to_details


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``total`` pairs for ``rank``; sizes differ by at most 1."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatParams:
    """Re-homes a module's parameters and gradients into two flat fp32 buffers, so the
    gradient exchange is ONE collective and clip/Adam touch one tensor.  ``extra`` floats
    are appended to the gradient buffer for piggy-backed statistics."""

    def __init__(self, module: nn.Module, extra: int = 8):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev, dt = params[0].device, params[0].dtype
        n = sum(p.numel() for p in params)
        self.numel = n
        self.flat = torch.empty(n, dtype=dt, device=dev)
        self.buf = torch.zeros(n + extra, dtype=dt, device=dev)
        self.flat_grad = self.buf[:n]
        self.extra = self.buf[n:]
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.flat_grad[off:off + k].view_as(p)
            off += k
        self.params = params
        self.flat_param = nn.Parameter(self.flat, requires_grad=True)
        self.flat_param.grad = self.flat_grad

    def zero_grad(self) -> None:
        self.buf.zero_()


class TrainStep:
    def __init__(self, model: nn.Module, features: torch.Tensor, edge_index: torch.Tensor,
                 learning_rate: float = 1e-3, weight_decay: float = 0.0,
                 gradient_clip: float = 1.0, scale_factor: float = 5.0,
                 process_group: Optional[dist.ProcessGroup] = None,
                 loss_fn: Optional[nn.Module] = None):
        self.model = model
        self.features = features
        self.edge_index = edge_index
        self.gradient_clip = gradient_clip
        self.loss_fn = loss_fn if loss_fn is not None else SkipGramLoss(scale_factor)
        self.flat = FlatParams(model)
        self.optimizer = torch.optim.Adam([self.flat.flat_param], lr=learning_rate,
                                          weight_decay=weight_decay)
        self.distributed = dist.is_available() and dist.is_initialized()
        self.group = process_group
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.rank = dist.get_rank(process_group) if self.distributed else 0

    def step(self, targets: torch.Tensor, contexts: torch.Tensor, negatives: torch.Tensor,
             global_batch: Optional[int] = None) -> torch.Tensor:
        """One optimisation step on this rank's shard.  ``global_batch`` = pairs of the whole
        step over all ranks (defaults to shard size x world).  Returns the device stats[8]
        tensor of the global batch (after the all-reduce); no host sync."""
        if global_batch is None:
            global_batch = targets.numel() * self.world
        self.model.train()
        self.flat.zero_grad()
        emb = self.model(self.features, self.edge_index)
        loss, stats = self.loss_fn.forward_stats(emb, targets, contexts, negatives,
                                                 global_batch=global_batch)
        loss.backward()
        self.flat.extra.copy_(stats)
        if self.world > 1:
            dist.all_reduce(self.flat.buf, op=dist.ReduceOp.SUM, group=self.group)
        if self.gradient_clip > 0:
            torch.nn.utils.clip_grad_norm_([self.flat.flat_param], self.gradient_clip)
        self.optimizer.step()
        return self.flat.extra

    def step_with_details(self, targets, contexts, negatives,
                          global_batch: Optional[int] = None) -> Dict[str, float]:
        return stats_to_details(self.step(targets, contexts, negatives, global_batch).clone())

    @torch.no_grad()
    def get_embeddings(self) -> torch.Tensor:
        """Counterpart of ``GraphSAGETrainer.get_embeddings`` (trainer.py:366-376)."""
        self.model.eval()
        return self.model(self.features, self.edge_index)
