"""End to end on one GPU: synthetic power-law memory graph -> device random walks -> co-occurrence
pairs -> GraphSAGETrainer (this package's drop-in for the reference trainer) -> embeddings.
Prints the loss per epoch and the neighbour-similarity gap the reference uses as its quality
number (`src/utils/metrics.py:17-76`: mean cosine of linked pairs minus mean cosine of random
pairs)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from graphsage_amd import (CooccurrencePairSampler, DegreeBiasedNegativeSampler, GraphSAGE,  # noqa: E402
                           GraphSAGETrainer, RandomWalkGenerator)
from graphsage_amd.synth import memory_features, powerlaw_graph  # noqa: E402


def similarity_gap(emb, ei, samples=200_000, seed=0):
    g = torch.Generator(device=emb.device).manual_seed(seed)
    e = torch.randint(0, ei.size(1), (samples,), device=emb.device, generator=g)
    linked = (emb[ei[0, e]] * emb[ei[1, e]]).sum(1).mean()
    a = torch.randint(0, emb.size(0), (samples,), device=emb.device, generator=g)
    b = torch.randint(0, emb.size(0), (samples,), device=emb.device, generator=g)
    rand = (emb[a] * emb[b]).sum(1).mean()
    return float(linked), float(rand)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=100_000)
    ap.add_argument("--avg-degree", type=int, default=20)
    ap.add_argument("--walk-length", type=int, default=20)
    ap.add_argument("--walks-per-node", type=int, default=4)
    ap.add_argument("--window", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=1 << 23, help="pairs sampled from the walks")
    ap.add_argument("--batch-size", type=int, default=1 << 16)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--lr", type=float, default=0.003)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)

    t0 = time.time()
    ei = powerlaw_graph(a.nodes, a.avg_degree, seed=42, device=dev)
    x = memory_features(a.nodes, seed=42, device=dev)
    walks, lengths = RandomWalkGenerator(ei, a.nodes, a.walk_length, a.walks_per_node, seed=42).generate_all_walks()
    sampler = CooccurrencePairSampler(a.window)
    offsets = sampler.index(walks, lengths)
    targets, contexts = sampler.sample_pairs(walks, lengths, offsets, a.pairs)
    neg = DegreeBiasedNegativeSampler(ei.cpu(), a.nodes)
    torch.cuda.synchronize()
    print(f"graph N={a.nodes} E={ei.size(1)}; {lengths.numel()} walks define {int(offsets[-1])} pairs, "
          f"{a.pairs} sampled; prepared in {time.time() - t0:.1f}s", flush=True)

    model = GraphSAGE(7, 64, 64, 2, 0.3).to(dev)
    model.eval()
    with torch.no_grad():
        before = similarity_gap(model(x, ei), ei)
    cfg = {"training": {"learning_rate": a.lr, "batch_size": a.batch_size, "gradient_clip": 1.0,
                        "early_stopping_patience": a.epochs + 1},
           "negatives": {"per_positive": 5}, "validation": {"val_fraction": 0.05}}
    trainer = GraphSAGETrainer(model, x, ei, (targets, contexts), neg, cfg, verbose=True)
    t0 = time.time()
    trainer.train(a.epochs)
    torch.cuda.synchronize()
    secs = time.time() - t0
    after = similarity_gap(trainer.get_embeddings(), ei)
    steps = sum(1 for _ in range(a.epochs)) * len(trainer.batch_generator)
    print(json.dumps({
        "epochs": a.epochs, "steps": steps, "seconds": round(secs, 2),
        "pairs_per_s": round(a.epochs * trainer.train_pairs[0].numel() / secs),
        "train_loss_first_last": [trainer.history[0]["train"]["loss"], trainer.history[-1]["train"]["loss"]],
        "val_loss_first_last": [trainer.history[0]["val"].get("loss"), trainer.history[-1]["val"].get("loss")],
        "cos_linked_minus_random_before": round(before[0] - before[1], 4),
        "cos_linked_minus_random_after": round(after[0] - after[1], 4),
    }))


if __name__ == "__main__":
    main()
