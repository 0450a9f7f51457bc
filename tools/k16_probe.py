"""The input layer's projection (project_fwd_k16_k) at C4 / C5 size: python tools/k16_probe.py [N] [Fo ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from graphsage_amd import _lib, ops


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    for fo in [int(a) for a in sys.argv[2:]] or [64, 256]:
        g = torch.Generator(device="cuda").manual_seed(fo)
        agg = torch.rand(n, 8, device="cuda", generator=g)
        x = torch.rand(n, 8, device="cuda", generator=g)
        wl = torch.randn(fo, 8, device="cuda", generator=g)
        wr = torch.randn(fo, 8, device="cuda", generator=g)
        b = torch.randn(fo, device="cuda", generator=g)
        t_eval = timed(lambda: ops.project_fwd(agg, x, wl, b, wr, epilogue=_lib.EPI_RELU))
        line = f"N={n} Fo={fo}: relu {t_eval:.3f} ms"
        if fo == 64:
            t_tr = timed(lambda: ops.project_fwd(agg, x, wl, b, wr, epilogue=_lib.EPI_RELU_DROPOUT, p=0.3, seed=3,
                                                 want_relu_sign=True))
            line += f", relu + dropout + sign bits {t_tr:.3f} ms"
        print(line, flush=True)


if __name__ == "__main__":
    main()
