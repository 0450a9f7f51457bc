"""Drop-in ``SkipGramLoss`` for the reference's ``src/model/loss.py:20-195``.

``forward`` and ``forward_with_details`` keep the reference signatures and the 7-key
details dict (``loss.py:185-193``); the whole loss is one fused K4 launch and the details
cost ONE device-to-host copy instead of seven ``.item()`` syncs.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .ops import SkipGramFunction

DETAIL_KEYS = ("total_loss", "pos_loss", "neg_loss", "pos_score_mean", "neg_score_mean",
               "pos_prob", "neg_prob")


class SkipGramLoss(nn.Module):
    """L = -mean_b log s(s*<e_t,e_c>) - mean_{b,k} log s(-s*<e_n,e_t>), s = scale_factor."""

    def __init__(self, scale_factor: float = 5.0, deterministic: Optional[bool] = None):
        """``deterministic``: True = sort-based backward (bitwise reproducible gradients),
        False = float-atomic scatter, None = follow GSAGE_LOSS_BWD (default auto: whichever is
        faster for the batch and table size, see ops.loss_bwd_deterministic)."""
        super().__init__()
        self.scale_factor = scale_factor
        self.deterministic = deterministic
        # set by an owner that consumes the embedding gradient itself and at once (TrainStep):
        # the gradient is then built in a persistent all-zero table (ops.GradTable)
        self.grad_table = None

    def _run(self, embeddings, targets, contexts, negatives, global_batch: Optional[int]):
        B = targets.numel()
        K = negatives.size(1) if negatives.dim() == 2 else 0
        gb = B if global_batch is None else int(global_batch)
        return SkipGramFunction.apply(embeddings, targets, contexts, negatives,
                                      float(self.scale_factor), float(gb),
                                      float(gb * max(K, 1)), self.deterministic, self.grad_table)

    def forward(self, embeddings: torch.Tensor, targets: torch.Tensor, contexts: torch.Tensor,
                negatives: torch.Tensor, global_batch: Optional[int] = None) -> torch.Tensor:
        """``global_batch``: total pairs of the step when this call sees one shard of it
        (data-parallel training); the shard losses then sum to the global mean."""
        loss, _ = self._run(embeddings, targets, contexts, negatives, global_batch)
        return loss

    def forward_stats(self, embeddings, targets, contexts, negatives,
                      global_batch: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(loss, device stats[8]) without any host sync (training loops that log rarely)."""
        return self._run(embeddings, targets, contexts, negatives, global_batch)

    def forward_with_details(self, embeddings, targets, contexts, negatives,
                             global_batch: Optional[int] = None
                             ) -> Tuple[torch.Tensor, Dict[str, float]]:
        loss, stats = self._run(embeddings, targets, contexts, negatives, global_batch)
        return loss, stats_to_details(stats)


    def release_grad_table(self) -> None:
        """Owner's call after the backward pass has consumed the embedding gradient."""
        if self.grad_table is not None:
            self.grad_table.release()


def stats_to_details(stats: torch.Tensor) -> Dict[str, float]:
    host = stats.detach().tolist()  # the single D2H copy
    if int(host[7]) & _lib.FLAG_INDEX_OUT_OF_RANGE:
        raise IndexError("targets/contexts/negatives hold node ids outside the embedding table")
    return {k: host[i] for i, k in enumerate(DETAIL_KEYS)}


def create_loss(loss_type: str = "skipgram", **kwargs) -> nn.Module:
    """Counterpart of ``create_loss`` (``loss.py:327-349``); only the loss the trainer uses
    (``trainer.py:156``) is on the hot path."""
    if loss_type == "skipgram":
        return SkipGramLoss(**kwargs)
    raise ValueError(f"Unknown loss type: {loss_type} (the MI355X hot path implements 'skipgram')")
