"""Transposed gather at C4 size with a zero-row mask of a given density: time per call against
the unmasked call (python tools/gather_mask_probe.py [density ...])."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from graphsage_amd.graph import build_csr
from graphsage_amd.ops import segment_gather
from graphsage_amd.synth import powerlaw_graph


def pack_bits(flags):
    n = flags.numel()
    f = torch.cat([flags.to(torch.int64), torch.zeros((-n) % 32, dtype=torch.int64, device=flags.device)])
    words = (f.view(-1, 32) << torch.arange(32, device=flags.device)).sum(1)
    return torch.where(words < 2**31, words, words - 2**32).to(torch.int32)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dens = [float(a) for a in sys.argv[1:]] or [0.0, 0.1, 0.3, 0.52, 0.8, 0.95, 1.0]
    n = 10_000_000
    g = build_csr(powerlaw_graph(n, 20, seed=42, device="cuda"), n)
    x = torch.randn(n, 64, device="cuda")
    base = torch.zeros(n, 64, device="cuda")
    kw = dict(long_rows=g.long_rows_t, n_long=g.n_long[1:2])
    t0 = timed(lambda: segment_gather(g.rowptr_t, g.col_t, x, add_to=base, out=base, **kw))
    print(f"unmasked           {t0:7.3f} ms", flush=True)
    for d in dens:
        live = torch.rand(n, device="cuda") < d
        m = pack_bits(live)
        t = timed(lambda: segment_gather(g.rowptr_t, g.col_t, x, add_to=base, out=base, src_nz_mask=m, **kw))
        print(f"density {d:4.2f}       {t:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
