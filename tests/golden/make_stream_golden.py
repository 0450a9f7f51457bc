"""stream_golden.npz: what the REFERENCE's StreamingBatchGenerator
(src/training/batch_generator.py:205-309, loaded by file path: its package __init__ pulls in
torch_geometric) yields for a set of ragged walks -- the shuffled walk order it used, every
batch's targets and contexts, the batch sizes and its pair estimate.  Only this data file
travels; the reference never does.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_stream_golden.py
"""
import importlib.util
import os
import random
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True

spec = importlib.util.spec_from_file_location("ref_batch_generator", f"{REF}/src/training/batch_generator.py")
bg = importlib.util.module_from_spec(spec)
sys.modules["ref_batch_generator"] = bg
spec.loader.exec_module(bg)


class NoNegatives:
    def sample(self, n, k):
        return torch.zeros(n, k, dtype=torch.long)


rng = random.Random(3)
walks = [[rng.randrange(50) for _ in range(rng.choice([2, 3, 5, 8, 13, 20]))] for _ in range(37)]
out = {"walk_len": np.array([len(w) for w in walks], dtype=np.int64),
       "walk_flat": np.array([v for w in walks for v in w], dtype=np.int64)}
for tag, window, bs, seed in (("a", 2, 16, 7), ("b", 10, 64, 8), ("c", 3, 1000, 9)):
    gen = bg.StreamingBatchGenerator(walks, window, NoNegatives(), batch_size=bs, num_negatives=2)
    random.seed(seed)
    order = list(range(len(walks)))
    random.shuffle(order)                        # the shuffle __iter__ will make (:259-261)
    random.seed(seed)
    batches = list(gen)
    out[f"{tag}.window"], out[f"{tag}.batch_size"] = np.int64(window), np.int64(bs)
    out[f"{tag}.order"] = np.array(order, dtype=np.int64)
    out[f"{tag}.sizes"] = np.array([b.size for b in batches], dtype=np.int64)
    out[f"{tag}.targets"] = torch.cat([b.targets for b in batches]).numpy()
    out[f"{tag}.contexts"] = torch.cat([b.contexts for b in batches]).numpy()
    out[f"{tag}.estimated_pairs"] = np.int64(gen.estimated_pairs)
    print(tag, "batches", len(batches), "pairs", int(out[f"{tag}.sizes"].sum()), "estimate", gen.estimated_pairs)
np.savez_compressed(os.path.join(OUT, "stream_golden.npz"), **out)
