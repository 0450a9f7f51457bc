"""Accumulation drift of the fused 64 -> 64 backward's weight gradients (long MFMA chains per
workgroup) on same-sign data at full size, against float64.
    PYTHONPATH=. python tools/bwd64_accuracy_probe.py [rows]"""
import sys

import torch

from graphsage_amd import _lib, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
for name, same in (("random signs", False), ("same sign", True)):
    mk = (lambda *s: torch.rand(*s, device=dev, generator=g)) if same else \
        (lambda *s: torch.randn(*s, device=dev, generator=g))
    agg, x, go = mk(n, 64), mk(n, 64), mk(n, 64)
    wl, wr = torch.randn(64, 64, device=dev, generator=g) / 8, torch.randn(64, 64, device=dev, generator=g) / 8
    rows = torch.nonzero(torch.rand(n, device=dev, generator=g) < 0.45).flatten().int()
    cap = rows.numel()
    live = ops.LiveRows(None, rows, torch.tensor([cap], dtype=torch.int32, device=dev), cap, 0, n)
    scale = torch.ones(n, device=dev)
    ga, gx, gwl, gb, gwr = ops.project_bwd_rows(go, None, None, agg, x, wl, wr, scale, _lib.EPI_NONE, 0.0, live)
    r = rows.long()
    d = go[r].double()
    for nm, got, want in (("W_l", gwl, d.t() @ agg[r].double()), ("b", gb, d.sum(0)), ("W_r", gwr, d.t() @ x[r].double())):
        err = (got.double() - want).abs().max().item()
        print(f"{name:13s} {nm:3s}: max |err| / max |grad| = {err / want.abs().max().item():.2e}")
