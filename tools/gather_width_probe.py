"""Plain mean aggregation at C4/C5 size for several feature widths: ms per launch and algorithmic
TB/s (python tools/gather_width_probe.py [F ...])."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from graphsage_amd.graph import build_csr
from graphsage_amd.ops import aggregate_mean
from graphsage_amd.synth import powerlaw_graph


def main():
    widths = [int(a) for a in sys.argv[1:]] or [8, 64, 128, 256]
    n = int(os.environ.get("PROBE_NODES", 10_000_000))
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    E = ei.size(1)
    g = build_csr(ei, n)
    del ei
    for f in widths:
        x = torch.randn(n, f, device="cuda")
        aggregate_mean(g, x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            aggregate_mean(g, x)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        alg = 4 * (n + 1) + 4 * E + 4 * E * f + 4 * n * f
        print(f"F={f:4d}  {ms:8.3f} ms  {alg / ms / 1e9:6.2f} TB/s algorithmic", flush=True)
        del x


if __name__ == "__main__":
    main()
