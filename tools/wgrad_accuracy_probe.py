"""Accuracy of gsage_hidden1_wgrad at full size against float64 (on the GPU, edges in chunks):
random-sign gradients and the worst case for accumulation bias, same-sign gradients.
    PYTHONPATH=. python tools/wgrad_accuracy_probe.py [nodes]"""
import sys

import torch

from graphsage_amd import _lib, default_cache, ops
from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
ei = powerlaw_graph(nodes, 20, seed=42, device=dev)
x = memory_features(nodes, seed=42, device=dev)
graph = default_cache().get(ei, nodes)
x8 = default_cache().padded(x, 8, keep=True)
g = torch.Generator(device=dev).manual_seed(5)
wl, wr = torch.randn(64, 8, device=dev, generator=g) / 3, torch.randn(64, 8, device=dev, generator=g) / 3
bl = torch.randn(64, device=dev, generator=g) / 3
p = 0.3
agg1 = ops.aggregate_mean(graph, x8)
box = {}
h1, _ = ops.project_fwd(agg1, x8, wl, bl, wr, epilogue=_lib.EPI_RELU_DROPOUT, seed=77, p=p, u_out=box)
t, c, neg = cooccurrence_batch(ei, nodes, 1 << 20, 5, seed=1001)
live = ops.touched_rows(t, c, neg, nodes)
n_live = int(live.n_rows.item())
rows = live.rows[:n_live].long()
print(f"N={nodes} E={ei.size(1)} live rows {n_live}")
is_live = torch.zeros(nodes, dtype=torch.bool, device=dev)
is_live[rows] = True
src, dst = ei[0], ei[1]
sel = is_live[dst]
src, dst = src[sel], dst[sel]
print(f"live edges {src.numel()}")
for name, same_sign in (("random signs", False), ("same sign", True)):
    ga = torch.zeros(nodes, 64, device=dev)
    gx = torch.zeros(nodes, 64, device=dev)
    r1 = torch.randn(n_live, 64, device=dev, generator=g)
    r2 = torch.randn(n_live, 64, device=dev, generator=g)
    if same_sign:
        r1, r2 = r1.abs(), r2.abs()
    ga[rows] = r1 * graph.inv_deg[rows, None]          # prescaled, as gsage_project_bwd_rows leaves it
    gx[rows] = r2
    gmax = torch.maximum(ga.abs().max(), gx.abs().max()).reshape(1).view(torch.int32)
    gwl, gb, gwr = ops.hidden1_wgrad(graph, box["u"], ga, gx, gmax, live, _lib.EPI_RELU_DROPOUT, p)
    gh = torch.zeros(nodes, 64, dtype=torch.float64, device=dev)
    step = 10_000_000
    for lo in range(0, src.numel(), step):
        gh.index_add_(0, src[lo:lo + step], ga[dst[lo:lo + step]].double())
    gh[rows] += gx[rows].double()
    dpre = gh * (h1 > 0) / (1.0 - p)
    del gh
    ewl = dpre.t() @ agg1.double()
    eb = dpre.sum(0)
    ewr = dpre.t() @ x8.double()
    del dpre
    for nm, got, want in (("W_l", gwl, ewl), ("b", gb, eb), ("W_r", gwr, ewr)):
        err = (got.double() - want).abs().max().item()
        print(f"{name:13s} {nm:3s}: max |err| {err:.3e}  / max |grad| {want.abs().max().item():.3e} = "
              f"{err / want.abs().max().item():.2e}")
