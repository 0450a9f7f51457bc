"""The 256 -> 256 projection of C5 as the encoder calls it: column-blocked operands, the range of the
activations handed in (no absmax passes), ReLU and L2-norm epilogues; ms per call by HIP events
and the error against float64 on the first rows.
    python tools/half_gemm_probe.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from graphsage_amd import _lib
from graphsage_amd.ops import project_fwd_blocked


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    f = 256
    torch.manual_seed(0)
    agg = torch.randn(f // 64, n, 64, device="cuda")
    x = torch.randn(f // 64, n, 64, device="cuda")
    wl = torch.randn(f, f, device="cuda") / f ** 0.5
    wr = torch.randn(f, f, device="cuda") / f ** 0.5
    b = torch.randn(f, device="cuda")
    amax = torch.maximum(agg.abs().max(), x.abs().max()).reshape(1).view(torch.int32)
    a64 = agg[:, :4096].permute(1, 0, 2).reshape(4096, f).double()
    x64 = x[:, :4096].permute(1, 0, 2).reshape(4096, f).double()
    pre = a64 @ wl.double().T + b.double() + x64 @ wr.double().T
    for epi, out_blocked, name in ((_lib.EPI_RELU, True, "relu, blocked out"), (_lib.EPI_L2NORM, False, "l2norm")):
        project_fwd_blocked(agg, x, wl, b, wr, epi, True, out_blocked, amax_in=amax)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out, _ = project_fwd_blocked(agg, x, wl, b, wr, epi, True, out_blocked, amax_in=amax)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        got = out[:, :4096].permute(1, 0, 2).reshape(4096, f).double() if out_blocked else out[:4096].double()
        ref = torch.relu(pre) if epi == _lib.EPI_RELU else torch.nn.functional.normalize(pre, dim=1)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        gb = 4.0 * n * 3 * f / 1e9
        print(f"N={n} 256->256 {name}: {ms:7.3f} ms  {gb / ms:6.2f} TB/s of its streams  "
              f"{3 * 2 * n * 2 * f * f / ms / 1e9:6.0f} TF/s issued  err {err:.2e}  "
              f"", flush=True)


if __name__ == "__main__":
    main()
