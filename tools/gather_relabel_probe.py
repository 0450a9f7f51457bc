"""does clustering the hot SOURCE rows help the narrow (F = 8) gather?
The destination order stays as it is (so nothing has to be undone at the output); only the
source table is permuted (descending out-degree first) and `col` relabelled through the inverse
permutation.  Prints ms per launch for F = 8 and F = 64, original vs relabelled, on the C4 graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from graphsage_amd import default_cache, ops            # noqa: E402
from graphsage_amd.synth import powerlaw_graph          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ei = powerlaw_graph(n, 20, seed=42, device="cuda")
g = default_cache().get(ei, n)
deg = g.out_degree()


def timed(col, x, reps=5):
    out = torch.empty(n, x.size(1), device="cuda")
    for _ in range(2):
        ops.segment_gather(g.rowptr, col, x, row_scale=g.inv_deg, out=out, long_rows=g.long_rows, n_long=g.n_long[0:1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.segment_gather(g.rowptr, col, x, row_scale=g.inv_deg, out=out, long_rows=g.long_rows, n_long=g.n_long[0:1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for kind in ("degree", "bucket"):
    if kind == "degree":
        perm = torch.argsort(deg, descending=True, stable=True)          # new position -> old id
    else:
        # coarse: hot rows (top 1/16 by degree) first, everything else in the original order
        thr = torch.quantile(deg[torch.randint(0, n, (1 << 20,), device="cuda")].float(), 1 - 1 / 16)
        hot = deg.float() > thr
        perm = torch.cat([hot.nonzero().flatten(), (~hot).nonzero().flatten()])
    rank = torch.empty_like(perm)
    rank[perm] = torch.arange(n, device="cuda")
    col2 = rank[g.col.long()].to(torch.int32)
    for F in (8, 64):
        x = torch.rand(n, F, device="cuda")
        t_a, out_a = timed(g.col, x)
        x2 = x[perm].contiguous()
        t_b, out_b = timed(col2, x2)
        same = torch.equal(out_a, out_b)
        print(f"N={n} F={F} relabel={kind}: original {t_a:.3f} ms, relabelled sources {t_b:.3f} ms "
              f"({(1 - t_b / t_a) * 100:+.1f} % time), identical output: {same}", flush=True)
        del x, x2, out_a, out_b
    del col2, rank, perm
