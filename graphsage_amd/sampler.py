"""Negative samplers: counterparts of ``src/walks/negative_sampler.py``.

``DegreeBiasedNegativeSampler`` keeps the reference's constructor, attributes (``probs``,
``prob_table``, ``alias``) and ``sample(num_samples, num_negatives)``.  What changes is where
the time goes: the out-degree comes from one ``bincount`` instead of a Python loop over every
edge (``negative_sampler.py:76-79``), and the Vose alias table is built by native host code
(``gsage_alias_build_host``) instead of an interpreter loop over every node (``:113-141``) --
bit-identical tables, ~1000x faster, which is what makes a 10 M-node sampler usable at all.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib


class DegreeBiasedNegativeSampler:
    """P(v) proportional to out_degree(v)^exponent (Word2Vec's 0.75), sampled by the alias method."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, exponent: float = 0.75,
                 device: Optional[torch.device] = None):
        self.num_nodes = num_nodes
        self.exponent = exponent
        self.device = device or torch.device("cpu")
        if edge_index.numel() == 0:
            degree = torch.ones(num_nodes, dtype=torch.float32)
        else:
            # out-degree = multiplicity of each id in edge_index[0]; exact in float32 below 2^24
            degree = torch.bincount(edge_index[0], minlength=num_nodes).to("cpu", torch.float32)
            degree = degree + 1e-8        # isolated nodes keep a non-zero probability (:82)
        # the few O(N) float32 ops run on the host with the same torch ops as the reference,
        # so the table is bit-identical to the reference's
        weights = degree ** exponent
        probs = weights / weights.sum()
        self.probs = probs.to(self.device)
        self._seed, self._draws = None, 0
        self._setup_alias_table(probs)

    def _setup_alias_table(self, probs_cpu: torch.Tensor) -> None:
        n = self.num_nodes
        scaled = np.ascontiguousarray(probs_cpu.numpy() * n, dtype=np.float32)
        prob_table = np.empty(n, dtype=np.float32)
        alias = np.empty(n, dtype=np.int64)
        rc = _lib.load().gsage_alias_build_host(scaled.ctypes.data, n, prob_table.ctypes.data,
                                                alias.ctypes.data)
        _lib.check(rc, "gsage_alias_build_host")
        self.alias = torch.from_numpy(alias).to(self.device)
        self.prob_table = torch.from_numpy(prob_table).to(self.device)

    def sample(self, num_samples: int, num_negatives: int = 5) -> torch.Tensor:
        """int64 [num_samples, num_negatives]; same draw rule as the reference
        (``negative_sampler.py:147-178``: uniform bin, uniform threshold, alias or bin).

        On a GPU the whole call is ONE kernel (``gsage_alias_sample``: a counter-hash stream
        seeded once from torch's CPU generator, so ``torch.manual_seed`` reproduces a run) and the
        library must be present.  A sampler left on the CPU -- the reference's default device --
        draws with the reference's own three torch ops."""
        total = num_samples * num_negatives
        if torch.device(self.device).type == "cuda":
            return self._sample_device(total).view(num_samples, num_negatives)
        idx = torch.randint(0, self.num_nodes, (total,), device=self.device)
        u = torch.rand(total, device=self.device)
        samples = torch.where(u >= self.prob_table[idx], self.alias[idx], idx)
        return samples.view(num_samples, num_negatives)

    def _sample_device(self, total: int) -> torch.Tensor:
        lib = _lib.load()
        if getattr(self, "_seed", None) is None:
            self._seed = int(torch.empty((), dtype=torch.int64).random_().item())
            self._draws = 0
        dev = self.prob_table.device
        with torch.cuda.device(dev):
            out = torch.empty(total, dtype=torch.int64, device=dev)
            rc = lib.gsage_alias_sample(self.prob_table.data_ptr(), self.alias.data_ptr(),
                                        self.num_nodes, total, self._seed & 0xFFFFFFFFFFFFFFFF,
                                        self._draws, out.data_ptr(), _lib.stream_handle())
        _lib.check(rc, "gsage_alias_sample")
        self._draws += total
        return out

    def reseed(self, seed: int) -> None:
        """Restart the device draw stream at counter 0 of ``seed`` (tests, reproducible runs)."""
        self._seed, self._draws = int(seed), 0

    def state_dict(self) -> dict:
        """Seed and position of the device draw stream (checkpoints: a resumed run continues the
        stream instead of starting another)."""
        return {"seed": self._seed, "draws": int(self._draws)}

    def load_state_dict(self, sd: dict) -> None:
        self._seed = None if sd.get("seed") is None else int(sd["seed"])
        self._draws = int(sd.get("draws", 0))

    def get_probability(self, node_idx: int) -> float:
        return self.probs[node_idx].item()

    def to(self, device: torch.device) -> "DegreeBiasedNegativeSampler":
        self.device = device
        self.probs = self.probs.to(device)
        self.alias = self.alias.to(device)
        self.prob_table = self.prob_table.to(device)
        return self


class UniformNegativeSampler:
    """Counterpart of ``UniformNegativeSampler`` (``negative_sampler.py:278-316``)."""

    def __init__(self, num_nodes: int, device: Optional[torch.device] = None):
        self.num_nodes = num_nodes
        self.device = device or torch.device("cpu")

    def sample(self, num_samples: int, num_negatives: int = 5) -> torch.Tensor:
        return torch.randint(0, self.num_nodes, (num_samples, num_negatives), device=self.device)
