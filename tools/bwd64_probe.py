"""Fused 64 -> 64 backward at C4 size (L2-norm epilogue, half of the gradient rows zero):
ms per call with and without the zero-row mask output (python tools/bwd64_probe.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from graphsage_amd import _lib
from graphsage_amd.ops import project_bwd, project_fwd


def main():
    n = 10_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    agg = torch.randn(n, 64, device="cuda", generator=g)
    x = torch.randn(n, 64, device="cuda", generator=g)
    wl = torch.randn(64, 64, device="cuda", generator=g) / 8
    wr = torch.randn(64, 64, device="cuda", generator=g) / 8
    b = torch.randn(64, device="cuda", generator=g)
    scale = torch.rand(n, device="cuda", generator=g)
    out, rnorm = project_fwd(agg, x, wl, b, wr, epilogue=_lib.EPI_L2NORM)
    gout = torch.randn(n, 64, device="cuda", generator=g)
    gout *= (torch.rand(n, 1, device="cuda", generator=g) < 0.5)
    for want in (False, True, False, True):
        fn = lambda: project_bwd(gout, out, rnorm, agg, x, wl, wr, scale, _lib.EPI_L2NORM, 0.0, True,
                                 want_nz_mask=want)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"zero-row mask {'on ' if want else 'off'}: {e0.elapsed_time(e1) / 10:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
