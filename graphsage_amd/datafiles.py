"""The preprocessed-data directory of the reference (``scripts/generate_data.py:175-212``, read
back by ``scripts/train.py:74-99``): ``features.pt`` (fp32 [N, 7]), ``edge_index.pt`` (int64
[2, E]), ``pairs.pt`` (``{'targets', 'contexts'}`` int64), ``metadata.json`` and, optionally,
``walks/walks.json``.  Same file names and contents, so either side can read the other's
directory.  Two differences in how they are read: tensors are loaded with ``weights_only=True``
(data files are data, never code), and the pairs stay two tensors instead of becoming a Python
list of 10^6..10^9 tuples (``train.py:88``), which is what the device-resident batch generator
takes anyway (the list form is accepted too).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

import torch


def save_training_data(data_dir: str, features: torch.Tensor, edge_index: torch.Tensor,
                       targets: torch.Tensor, contexts: torch.Tensor,
                       metadata: Optional[Dict] = None,
                       walks: Optional[List[List[int]]] = None) -> None:
    os.makedirs(data_dir, exist_ok=True)
    torch.save(features.detach().cpu().to(torch.float32), os.path.join(data_dir, "features.pt"))
    torch.save(edge_index.detach().cpu().to(torch.int64), os.path.join(data_dir, "edge_index.pt"))
    torch.save({"targets": targets.detach().cpu().to(torch.int64),
                "contexts": contexts.detach().cpu().to(torch.int64)},
               os.path.join(data_dir, "pairs.pt"))
    meta = {"num_nodes": int(features.size(0)), "num_edges": int(edge_index.size(1)),
            "num_pairs": int(targets.numel()), "feature_dim": int(features.size(1))}
    meta.update(metadata or {})
    with open(os.path.join(data_dir, "metadata.json"), "w") as fh:
        json.dump(meta, fh, indent=2)
    if walks is not None:
        os.makedirs(os.path.join(data_dir, "walks"), exist_ok=True)
        with open(os.path.join(data_dir, "walks", "walks.json"), "w") as fh:
            json.dump(walks, fh)


def load_training_data(data_dir: str, device="cuda"
                       ) -> Tuple[torch.Tensor, torch.Tensor, Tuple[torch.Tensor, torch.Tensor], Dict]:
    """(features, edge_index, (targets, contexts), metadata), tensors on ``device``."""
    dev = torch.device(device)
    features = torch.load(os.path.join(data_dir, "features.pt"), map_location="cpu", weights_only=True)
    edge_index = torch.load(os.path.join(data_dir, "edge_index.pt"), map_location="cpu", weights_only=True)
    pairs = torch.load(os.path.join(data_dir, "pairs.pt"), map_location="cpu", weights_only=True)
    if features.dim() != 2 or edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError("features.pt must be [N, F] and edge_index.pt [2, E]")
    targets, contexts = pairs["targets"].to(torch.int64), pairs["contexts"].to(torch.int64)
    if targets.shape != contexts.shape:
        raise ValueError("pairs.pt: targets and contexts differ in length")
    meta_path = os.path.join(data_dir, "metadata.json")
    metadata = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    return (features.to(dev, torch.float32), edge_index.to(dev, torch.int64),
            (targets.to(dev), contexts.to(dev)), metadata)


def load_walks(data_dir: str) -> List[List[int]]:
    with open(os.path.join(data_dir, "walks", "walks.json")) as fh:
        return json.load(fh)
