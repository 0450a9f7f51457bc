"""Times the device feeder (walks, pair index, per-batch pair sampling) on a synthetic graph."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from graphsage_amd import CooccurrencePairSampler, RandomWalkGenerator
from graphsage_amd.synth import powerlaw_graph


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--avg-degree", type=int, default=10)
    ap.add_argument("--walk-length", type=int, default=80)
    ap.add_argument("--walks-per-node", type=int, default=10)
    ap.add_argument("--window", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1 << 20)
    a = ap.parse_args()
    ei = powerlaw_graph(a.nodes, a.avg_degree, seed=42).cuda()
    gen = RandomWalkGenerator(ei, a.nodes, a.walk_length, a.walks_per_node, seed=42)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    ms_w, (walks, lengths) = timed(gen.generate_all_walks)
    s = CooccurrencePairSampler(a.window)
    ms_i, off = timed(lambda: s.index(walks, lengths))
    ms_s, _ = timed(lambda: s.sample_pairs(walks, lengths, off, a.batch), reps=10)
    steps = int(lengths.sum().item()) - lengths.numel()
    print(f"nodes={a.nodes} walks={lengths.numel()} steps={steps} total_pairs={int(off[-1])}")
    print(f"walks {ms_w:.2f} ms ({steps / ms_w / 1e6:.2f} G steps/s)  index {ms_i:.2f} ms  "
          f"sample {a.batch} pairs {ms_s:.3f} ms ({a.batch / ms_s / 1e3:.1f} M pairs/s)")


if __name__ == "__main__":
    main()
