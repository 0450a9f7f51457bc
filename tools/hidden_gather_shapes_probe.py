"""Does the size switch of the recomputed aggregation (GSAGE_HIDDEN1_MIN_BYTES = 96 MiB of stored
activation) hold on other degree distributions than the benchmark's (power law, gamma 2.5, mean
20)?  Plain gather over the stored activation against gsage_segment_gather_hidden1, 64 columns,
eval, for mean degree 5 / 20 / 60, tail exponent 2.1 / 2.5 / 3.5 and a capped (near-regular) graph,
at table sizes either side of the switch."""
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
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


fo = 64
g = torch.Generator(device="cuda").manual_seed(fo)
wl = torch.nn.functional.pad(torch.randn(fo, 7, device="cuda", generator=g) / 2, (0, 1)).contiguous()
wr = torch.nn.functional.pad(torch.randn(fo, 7, device="cuda", generator=g) / 2, (0, 1)).contiguous()
b = torch.randn(fo, device="cuda", generator=g) / 4
shapes = [("power law 2.5, mean 20 (benchmark)", 20, 2.5, None), ("power law 2.5, mean 5", 5, 2.5, None),
          ("power law 2.5, mean 60", 60, 2.5, None), ("power law 2.1, mean 20 (heavier tail)", 20, 2.1, None),
          ("power law 3.5, mean 20 (lighter tail)", 20, 3.5, None), ("degrees capped at 40, mean 20", 20, 2.5, 40)]
for name, deg, gamma, cap in shapes:
    line = []
    for n in (200_000, 400_000, 800_000, 3_000_000):
        ei = powerlaw_graph(n, deg, gamma=gamma, seed=42, device="cuda", max_degree=cap)
        x8 = torch.nn.functional.pad(memory_features(n, seed=42, device="cuda"), (0, 1)).contiguous()
        graph = default_cache().get(ei, n)
        agg1 = ops.aggregate_mean(graph, x8)
        h1, _ = ops.project_fwd(agg1, x8, wl, b, wr, epilogue=_lib.EPI_RELU)
        tp = timed(lambda: ops.aggregate_mean(graph, h1))
        tn = timed(lambda: ops.aggregate_hidden1(graph, agg1, x8, wl, b, wr, _lib.EPI_RELU))
        line.append(f"{n // 1000}k ({n * fo * 4 >> 20} MiB): {tp:.3f} / {tn:.3f}")
        del ei, graph, h1, agg1
        default_cache().clear() if hasattr(default_cache(), "clear") else None
    print(f"{name}: plain / recomputed ms   " + "   ".join(line), flush=True)
