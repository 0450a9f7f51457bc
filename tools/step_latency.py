#!/usr/bin/env python
"""Side measurements quoted in DESIGN.md (not the bench line):
  * reference-semantics step (B = 1024) on the C1 fixture and on C2, eager vs captured HIP graph;
  * full-graph inference forward, C5 shape (3 layers, 256 wide) and the default 2x64.
Usage: python tools/step_latency.py [--c5-nodes N]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from graphsage_amd import GraphSAGE, StructuralEncoder, TrainStep  # noqa: E402
from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph  # noqa: E402


def time_steps(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def train_latency(name, x, ei, n, B=1024, K=5):
    out = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(0)
        model = GraphSAGE(7, 64, 64, 2, 0.3).cuda()
        ts = TrainStep(model, x, ei)
        batches = [cooccurrence_batch(ei, n, B, K, seed=s) for s in range(8)]
        i = [0]
        if mode == "graph":
            ts.capture_step(B, K)

            def fn():
                ts.step_captured(*batches[i[0] % 8]); i[0] += 1
        else:
            def fn():
                ts.step(*batches[i[0] % 8]); i[0] += 1
        ms = time_steps(fn)
        out[mode] = {"ms_per_step": ms, "pairs_per_s": B / ms * 1e3}
    print(json.dumps({"workload": name, "B": B, **out}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c5-nodes", type=int, default=10_000_000)
    args = ap.parse_args()
    z = np.load(os.path.join(ROOT, "tests", "golden", "c1_graph.npz"))
    train_latency("C1 (reference default graph: 500 nodes, 1106 edges)",
                  torch.from_numpy(z["features"]).cuda(), torch.from_numpy(z["edge_index"]).cuda(), 500)
    n = 100_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    train_latency("C2 (100k nodes)", memory_features(n, device="cuda"), ei, n)
    for nodes, cfg in ((args.c5_nodes, dict(hidden_channels=256, out_channels=256, num_layers=3)),
                       (args.c5_nodes, dict(hidden_channels=64, out_channels=64, num_layers=2))):
        ei = powerlaw_graph(nodes, 20, seed=42, device="cuda")
        x = memory_features(nodes, device="cuda")
        torch.manual_seed(0)
        enc = StructuralEncoder(model=GraphSAGE(7, dropout=0.0, **cfg))
        r = enc.benchmark(x, ei, num_runs=5, warmup=2)
        r["model"] = cfg
        print(json.dumps(r))
        del enc, ei, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
