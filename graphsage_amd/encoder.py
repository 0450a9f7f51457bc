"""Inference entry: the in-scope part of ``StructuralEncoder`` (``src/inference/encoder.py``).

The reference's ``encode_all`` spends most of its time in host Python (view and feature
extraction, ``encoder.py:168-178``); the device work is three lines (``:181-185``): move the
tensors, run ``model(features, edge_index)`` in eval mode under ``no_grad``.  This class keeps
the checkpoint loading contract (``_load_model``, ``:107-141``: rebuild ``GraphSAGE`` from
``checkpoint['config']`` with ``dropout=0.0``, strict ``load_state_dict``) and that forward.
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch

from .graphsage import GraphSAGE


def model_from_checkpoint(checkpoint: dict, fallback_config: Optional[dict] = None) -> GraphSAGE:
    cfg = checkpoint["config"] if "config" in checkpoint else (fallback_config or {})
    mc, fc = cfg.get("model", {}), cfg.get("features", {})
    model = GraphSAGE(in_channels=fc.get("dimensions", 7), hidden_channels=mc.get("hidden_dim", 64),
                      out_channels=mc.get("output_dim", 64), num_layers=mc.get("num_layers", 2),
                      dropout=0.0, normalize_output=mc.get("normalize_output", True))
    model.load_state_dict(checkpoint["model_state_dict"])      # strict, PyG key layout
    return model


class StructuralEncoder:
    def __init__(self, model_path: Optional[str] = None, model: Optional[GraphSAGE] = None,
                 device: str = "cuda", config: Optional[Dict] = None):
        self.device = torch.device(device)
        self.config = config or {}
        if model is not None:
            self.model = model.to(self.device)
        elif model_path is not None:
            # weights_only: checkpoints are data, never code
            ckpt = torch.load(model_path, map_location=self.device, weights_only=True)
            self.model = model_from_checkpoint(ckpt, self.config).to(self.device)
        else:
            raise ValueError("Must provide either model_path or model")
        self.model.eval()
        self._inference_times = []

    @torch.no_grad()
    def encode_tensors(self, features: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        """``encoder.py:181-185``: fp32 [N, F] + int64 [2, E] -> fp32 [N, D] embeddings."""
        features = features.to(self.device)
        edge_index = edge_index.to(self.device)
        t0 = time.time()
        out = self.model(features, edge_index)
        self._inference_times.append(time.time() - t0)
        return out

    def get_embedding_dim(self) -> int:
        return self.model.out_channels

    def benchmark(self, features: torch.Tensor, edge_index: torch.Tensor, num_runs: int = 20,
                  warmup: int = 3) -> Dict[str, float]:
        """Protocol of ``scripts/benchmark_latency.py:66-105``: synchronise, time ``num_runs``
        forwards one by one, report mean / min / max milliseconds."""
        features, edge_index = features.to(self.device), edge_index.to(self.device)
        for _ in range(warmup):
            self.encode_tensors(features, edge_index)
        torch.cuda.synchronize(self.device)
        times = []
        for _ in range(num_runs):
            t0 = time.perf_counter()
            self.encode_tensors(features, edge_index)
            torch.cuda.synchronize(self.device)
            times.append((time.perf_counter() - t0) * 1e3)
        return {"mean_ms": sum(times) / len(times), "min_ms": min(times), "max_ms": max(times),
                "num_nodes": int(features.size(0)), "num_edges": int(edge_index.size(1))}
