"""The first hidden layer's aggregation at C4 / C5 size: plain gather over the stored activation
against gsage_segment_gather_hidden1 (python tools/hidden_gather_probe.py [N] [Fo ...])."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from graphsage_amd import _lib, ops
from graphsage_amd.graph import default_cache
from graphsage_amd.synth import memory_features, powerlaw_graph


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    widths = [int(a) for a in sys.argv[2:]] or [64, 256]
    md = os.environ.get("PROBE_MAX_DEGREE")
    ei = powerlaw_graph(n, 20, seed=42, device="cuda", max_degree=int(md) if md else None)
    x = memory_features(n, seed=42, device="cuda")
    x8 = torch.nn.functional.pad(x, (0, 1)).contiguous()
    graph = default_cache().get(ei, n)
    agg1 = ops.aggregate_mean(graph, x8)
    for fo in widths:
        g = torch.Generator(device="cuda").manual_seed(fo)
        wl = torch.nn.functional.pad(torch.randn(fo, 7, device="cuda", generator=g) / 2, (0, 1)).contiguous()
        wr = torch.nn.functional.pad(torch.randn(fo, 7, device="cuda", generator=g) / 2, (0, 1)).contiguous()
        b = torch.randn(fo, device="cuda", generator=g) / 4
        modes = [("eval", _lib.EPI_RELU, 0.0), ("train", _lib.EPI_RELU_DROPOUT, 0.3)]
        for name, epi, p in modes:
            if epi == _lib.EPI_RELU:
                h1, _ = ops.project_fwd(agg1, x8, wl, b, wr, epilogue=epi)
                sign = None
            else:
                h1, _, sign = ops.project_fwd(agg1, x8, wl, b, wr, epilogue=epi, p=p, seed=5, want_relu_sign=True)
            t_plain, want = timed(lambda: ops.aggregate_mean(graph, h1))
            t_new, got = timed(lambda: ops.aggregate_hidden1(graph, agg1, x8, wl, b, wr, epi, p=p, relu_sign=sign,
                                                             h1=None if sign is not None else h1))
            err = ((got - want).abs().max() / want.abs().max()).item()
            print(f"N={n} Fo={fo} {name}: plain gather {t_plain:7.3f} ms, recomputed {t_new:7.3f} ms, "
                  f"max |diff| / max |ref| = {err:.2e}", flush=True)
            del h1, want, got


if __name__ == "__main__":
    main()
