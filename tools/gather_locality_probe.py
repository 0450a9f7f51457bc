"""Same number of edges over ever smaller node tables (10 M / 1 M / 100 k nodes): what perfect
cache locality would buy the 8- and 64-wide gathers (quoted in DESIGN.md, K2)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from graphsage_amd import default_cache, ops
from graphsage_amd.synth import powerlaw_graph
for n, d in [(10_000_000, 20), (1_000_000, 200), (100_000, 2000)]:
    ei = powerlaw_graph(n, d, seed=1, device='cuda')
    g = default_cache().get(ei, n)
    for F in (8, 64):
        x = torch.rand(n, F, device='cuda')
        out = torch.empty_like(x)
        for _ in range(2): ops.segment_gather(g.rowptr, g.col, x, row_scale=g.inv_deg, out=out, long_rows=g.long_rows, n_long=g.n_long[0:1])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ops.segment_gather(g.rowptr, g.col, x, row_scale=g.inv_deg, out=out, long_rows=g.long_rows, n_long=g.n_long[0:1])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"N={n} E={ei.size(1)} F={F}: {ms:.3f} ms  {ei.size(1)/ms/1e6:.1f} G edges/s", flush=True)
    del ei, g, x, out
    default_cache().clear(); torch.cuda.empty_cache()
