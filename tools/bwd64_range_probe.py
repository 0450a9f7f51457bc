"""Row-list backward of the 64 -> 64 layer when rows differ by many orders of magnitude: the fp16 forms
scale dX per gradient row and dW per 8-row block, so what happens to a block whose rows are far apart?
Gradient rows and operand rows get independent magnitudes 2^U(-s, s); every output against float64.
    PYTHONPATH=. python tools/bwd64_range_probe.py [rows]         (GSAGE_BWD64=f32|dx for the other forms)"""
import os
import sys

import torch

from graphsage_amd import _lib, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
for spread in (0, 8, 16, 24, 40):
    def mk(s):
        mag = torch.exp2((torch.rand(n, 1, device=dev, generator=g) * 2 - 1) * s)
        return torch.randn(n, 64, device=dev, generator=g) * mag
    agg, x, go = mk(spread), mk(spread), mk(spread)
    wl, wr = torch.randn(64, 64, device=dev, generator=g) / 8, torch.randn(64, 64, device=dev, generator=g) / 8
    rows = torch.nonzero(torch.rand(n, device=dev, generator=g) < 0.45).flatten().int()
    cap = rows.numel()
    live = ops.LiveRows(None, rows, torch.tensor([cap], dtype=torch.int32, device=dev), cap, 0, n)
    scale = torch.ones(n, device=dev)
    ga, gx, gwl, gb, gwr = ops.project_bwd_rows(go, None, None, agg, x, wl, wr, scale, _lib.EPI_NONE, 0.0, live)
    r = rows.long()
    d = go[r].double()
    out = []
    for nm, got, want in (("W_l", gwl, d.t() @ agg[r].double()), ("b", gb, d.sum(0)), ("W_r", gwr, d.t() @ x[r].double())):
        out.append(f"{nm} {(got.double() - want).abs().max().item() / want.abs().max().item():.1e}")
    # dX row by row: every row against ITS OWN largest element
    for nm, got, w in (("d agg", ga, wl), ("d x", gx, wr)):
        want = d @ w.double()
        rel = ((got[r].double() - want).abs().max(1).values / want.abs().max(1).values.clamp_min(1e-300)).max().item()
        out.append(f"{nm} (row-wise) {rel:.1e}")
    print(f"[{os.environ.get('GSAGE_BWD64', 'default')}] rows 2^+-{spread}: " + "  ".join(out), flush=True)

# the adverse case: inside every block of 8 listed rows, the rows with the large gradients have the small
# operands and the other way round (all products of the same order, the two factors 2^r apart)
for r_ in (0, 4, 8, 12, 16, 20, 30):
    alt = (torch.arange(n, device=dev) & 1).float().mul(2).sub(1).unsqueeze(1)          # -1, +1, -1, ...
    up, down = torch.exp2(alt * (r_ / 2)), torch.exp2(-alt * (r_ / 2))
    go = torch.randn(n, 64, device=dev, generator=g) * up
    agg = torch.randn(n, 64, device=dev, generator=g) * down
    x = torch.randn(n, 64, device=dev, generator=g) * down
    rows = torch.arange(n, device=dev, dtype=torch.int32)
    live = ops.LiveRows(None, rows, torch.tensor([n], dtype=torch.int32, device=dev), n, 0, n)
    ga, gx, gwl, gb, gwr = ops.project_bwd_rows(go, None, None, agg, x, wl, wr, torch.ones(n, device=dev),
                                                _lib.EPI_NONE, 0.0, live)
    d = go.double()
    out = []
    for nm, got, want in (("W_l", gwl, d.t() @ agg.double()), ("W_r", gwr, d.t() @ x.double())):
        out.append(f"{nm} {(got.double() - want).abs().max().item() / want.abs().max().item():.1e}")
    print(f"[{os.environ.get('GSAGE_BWD64', 'default')}] adverse 2^{r_}: " + "  ".join(out), flush=True)
