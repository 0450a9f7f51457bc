import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from graphsage_amd import _lib, ops
from test_gpu_hidden_gather import _setup
n, deg, b, p = 5000, 20, 600, 0.3
ei, x8, wl, bias, wr, graph = _setup(n, 64, deg, seed=3 * n)
d = [t.cuda() for t in (x8, wl, bias, wr)]
agg1 = ops.aggregate_mean(graph, d[0])
h1, _, sign = ops.project_fwd(agg1, d[0], d[1], d[2], d[3], epilogue=_lib.EPI_RELU_DROPOUT, p=p, seed=11, want_relu_sign=True)
_, ws = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU_DROPOUT, p=p, relu_sign=sign, return_ws=True)
U = ws[: (n) * 64].view(torch.float32).view(n, 16)
lo = U[:, 7].contiguous().view(torch.int32).long() & 0xFFFFFFFF
hi = U[:, 15].contiguous().view(torch.int32).long() & 0xFFFFFFFF
m64 = lo | (hi << 32)
cols = torch.arange(64, device="cuda")
bitpos = 16 * (cols & 3) + (cols >> 2)
bits = ((m64[:, None] >> bitpos[None, :]) & 1).bool()
print("mask bits equal h1>0:", bool((bits == (h1 > 0)).all()), int((bits != (h1 > 0)).sum()))
g = torch.Generator().manual_seed(n)
t = torch.randint(0, n, (b,), generator=g).cuda(); c = torch.randint(0, n, (b,), generator=g).cuda(); neg = torch.randint(0, n, (b, 5), generator=g).cuda()
live = ops.touched_rows(t, c, neg, n)
rows = live.rows[: int(live.n_rows.item())].long()
deg_in = graph.in_degree()
for name, sel in (("all", rows), ("deg<=8", rows[deg_in[rows] <= 8]), ("deg>8", rows[deg_in[rows] > 8]), ("deg>64", rows[deg_in[rows] > 64])):
    gagg = torch.zeros(n, 64, device="cuda"); groot = torch.zeros(n, 64, device="cuda")
    gg = torch.Generator().manual_seed(1)
    gagg[sel] = torch.randn(sel.numel(), 64, generator=gg).cuda() * graph.inv_deg[sel, None]
    gwl, gb, gwr = ops.hidden1_dweights(graph, ws, gagg, groot, live, p)
    keep = (h1 > 0).double().cpu()
    gh = torch.zeros(n, 64, dtype=torch.float64)
    gh.index_add_(0, ei[0], gagg.double().cpu()[ei[1]])
    g1 = gh * keep / (1 - p)
    want = g1.sum(0)
    err = (gb.double().cpu() - want).abs()
    print(name, sel.numel(), "max err", float(err.max()), "at", int(err.argmax()), "scale", float(want.abs().max()))
    wantw = g1.T @ x8.double()
    print("   w_r err", float((gwr.double().cpu() - wantw).abs().max()), float(wantw.abs().max()))
