"""Forward projection of a wide layer at C5/C3 size: ms per call and fp32-equivalent TFLOP/s
(python tools/wide_gemm_probe.py [N] [F ...]); GSAGE_WIDE_GEMM=f32|split forces one kernel family."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from graphsage_amd import _lib
from graphsage_amd.ops import project_fwd


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    widths = [int(a) for a in sys.argv[2:]] or [256, 128]
    for f in widths:
        agg = torch.randn(n, f, device="cuda")
        x = torch.randn(n, f, device="cuda")
        wl = torch.randn(f, f, device="cuda") / f ** 0.5
        wr = torch.randn(f, f, device="cuda") / f ** 0.5
        b = torch.randn(f, device="cuda")
        for epi in (_lib.EPI_RELU, _lib.EPI_L2NORM):
            project_fwd(agg, x, wl, b, wr, epilogue=epi)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out, _ = project_fwd(agg, x, wl, b, wr, epilogue=epi)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            ref = torch.relu(agg[:4096].double() @ wl.double().T + b.double() + x[:4096].double() @ wr.double().T)
            if epi == _lib.EPI_L2NORM:
                pre = agg[:4096].double() @ wl.double().T + b.double() + x[:4096].double() @ wr.double().T
                ref = torch.nn.functional.normalize(pre, dim=1)
            err = ((out[:4096].double() - ref).abs().max() / ref.abs().max()).item()
            print(f"N={n} F={f} epi={epi}: {ms:7.3f} ms  {2 * n * 2 * f * f / ms / 1e9:6.1f} TFLOP/s  "
                  f"max err / max |ref| = {err:.2e}", flush=True)
        del agg, x


if __name__ == "__main__":
    main()
