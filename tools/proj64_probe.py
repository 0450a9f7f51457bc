"""proj64_fwd_k at C4 size (10 M rows, 64 -> 64, L2-norm epilogue): ms per launch and the result
checked against the generic tile kernel.  Used in round 2 to A/B two variants that were then
dropped: two waves per workgroup (2.03 vs 2.06 ms) and non-temporal loads/stores
(2.02 / 2.03 vs 2.05 ms).  The figures include the weight pack and the output allocation."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from graphsage_amd import _lib, ops   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = torch.Generator(device="cuda").manual_seed(0)
agg, x = torch.randn(n, 64, device="cuda", generator=g), torch.randn(n, 64, device="cuda", generator=g)
wl, wr = torch.randn(64, 64, device="cuda", generator=g) * 0.1, torch.randn(64, 64, device="cuda", generator=g) * 0.1
b = torch.randn(64, device="cuda", generator=g)
for epi in (_lib.EPI_L2NORM, _lib.EPI_RELU_DROPOUT):
    for _ in range(3):
        out, _ = ops.project_fwd(agg, x, wl, b, wr, epi, None, 7, 0.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out, _ = ops.project_fwd(agg, x, wl, b, wr, epi, None, 7, 0.3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    ref, _ = ops.project_fwd(agg[:200_000], x[:200_000], wl, b, wr, epi, None, 7, 0.3, force_generic=2)
    err = float((out[:200_000] - ref).abs().max())
    print(f"epilogue={epi}: {ms:.3f} ms per launch, "
          f"max |diff| vs generic tiles {err:.2e}", flush=True)
