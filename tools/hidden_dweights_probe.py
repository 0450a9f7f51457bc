"""The input layer's weight gradients at C4 size: masked transposed gather + first-layer dW kernel
against gsage_hidden1_dweights (python tools/hidden_dweights_probe.py [N] [B])."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from graphsage_amd import _lib, ops
from graphsage_amd.graph import default_cache
from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph


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
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    p = 0.3
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    x8 = torch.nn.functional.pad(x, (0, 1)).contiguous()
    graph = default_cache().get(ei, n)
    agg1 = ops.aggregate_mean(graph, x8)
    g = torch.Generator(device="cuda").manual_seed(1)
    wl = torch.nn.functional.pad(torch.randn(64, 7, device="cuda", generator=g) / 2, (0, 1)).contiguous()
    wr = torch.nn.functional.pad(torch.randn(64, 7, device="cuda", generator=g) / 2, (0, 1)).contiguous()
    bias = torch.randn(64, device="cuda", generator=g) / 4
    h1, _, sign = ops.project_fwd(agg1, x8, wl, bias, wr, epilogue=_lib.EPI_RELU_DROPOUT, p=p, seed=5,
                                  want_relu_sign=True)
    _, ws = ops.aggregate_hidden1(graph, agg1, x8, wl, bias, wr, _lib.EPI_RELU_DROPOUT, p=p, relu_sign=sign,
                                  return_ws=True)
    t, c, neg = cooccurrence_batch(ei, n, b, 5, seed=3)
    live = ops.touched_rows(t, c, neg, n)
    nl = int(live.n_rows.item())
    rows = live.rows[:nl].long()
    gagg = torch.zeros(n, 64, device="cuda")
    groot = torch.zeros(n, 64, device="cuda")
    gagg[rows] = torch.randn(nl, 64, device="cuda", generator=g) * graph.inv_deg[rows, None]
    groot[rows] = torch.randn(nl, 64, device="cuda", generator=g)
    live_edges = int(graph.in_degree()[rows].sum())

    def old():
        gh = ops.aggregate_mean_bwd(graph, gagg, prescaled=True, add_to=groot.clone(), live_mask=live.mask)
        return ops.project_bwd(gh, h1, None, agg1, x8, wl, wr, None, _lib.EPI_RELU_DROPOUT, p, False,
                               relu_sign=sign)[2:]

    t_clone, _ = timed(lambda: groot.clone())
    t_old, ref = timed(old)
    t_new, got = timed(lambda: ops.hidden1_dweights(graph, ws, gagg, groot, live, p))
    err = max(((a - r).abs().max() / r.abs().max()).item() for a, r in zip(got, ref))
    print(f"N={n} B={b}: {nl} live rows, {live_edges} live edges; gather + dW {t_old - t_clone:7.3f} ms "
          f"(clone of {t_clone:.3f} ms taken off), hidden1_dweights {t_new:7.3f} ms, "
          f"max |diff| / max |ref| = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
