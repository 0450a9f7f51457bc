"""Does the input aggregation (a request-rate bound gather) overlap with the rest of a training
step when it runs on a second stream?  Times: the step alone, the gather alone, both at once.
    python tools/overlap_probe.py [nodes]"""
import sys
import time

import torch

from graphsage_amd import GraphSAGE, TrainStep, default_cache, ops
from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
ei = powerlaw_graph(nodes, 20, seed=42, device=dev)
x = memory_features(nodes, seed=42, device=dev)
graph = default_cache().get(ei, nodes)
x8 = default_cache().padded(x, 8, keep=True)
torch.manual_seed(0)
model = GraphSAGE(7, 64, 64, 2, 0.3).to(dev)
step = TrainStep(model, x, ei)
B, K = 1 << 20, 5
batches = [cooccurrence_batch(ei, nodes, B, K, seed=1000 + i) for i in range(4)]
side = torch.cuda.Stream()
out = torch.empty(nodes, 8, device=dev)


def gather():
    ops.segment_gather(graph.rowptr, graph.col, x8, row_scale=graph.inv_deg, out=out,
                       long_rows=graph.long_rows, n_long=graph.n_long[0:1])


def timed(fn, n=10):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def both(i):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gather()
    step.step(*batches[i % 4], global_batch=B)
    torch.cuda.current_stream().wait_stream(side)


t_step = timed(lambda i: step.step(*batches[i % 4], global_batch=B))
t_gather = timed(lambda i: gather())
t_both = timed(both)
print(f"step alone {t_step:.3f} ms, gather alone {t_gather:.3f} ms, both at once {t_both:.3f} ms "
      f"(sum {t_step + t_gather:.3f}; hidden {t_step + t_gather - t_both:.3f} ms)")

# the same question against ONE streaming kernel (the 64 -> 64 projection, HBM-bandwidth bound)
from graphsage_amd import _lib
h = torch.randn(nodes, 64, device=dev)
a = torch.randn(nodes, 64, device=dev)
wl, wr, bl = torch.randn(64, 64, device=dev) / 8, torch.randn(64, 64, device=dev) / 8, torch.zeros(64, device=dev)


def proj(i):
    ops.project_fwd(a, h, wl, bl, wr, epilogue=_lib.EPI_L2NORM)


def proj_and_gather(i):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gather()
    proj(i)
    torch.cuda.current_stream().wait_stream(side)


t_p = timed(proj)
t_pg = timed(proj_and_gather)
print(f"projection alone {t_p:.3f} ms, gather alone {t_gather:.3f} ms, both at once {t_pg:.3f} ms "
      f"(hidden {t_p + t_gather - t_pg:.3f} ms)")
