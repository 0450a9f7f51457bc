"""Phase timing of gemm_half_k (a library built with -DHB_PROF): cycles of wave 0 of every
workgroup, summed per phase over the workgroup's chunks.  GSAGE_WIDE_GEMM=h python tools/probes/gemm_half_prof.py [N]"""
import ctypes as C
import sys
import numpy as np
import torch
from graphsage_amd import _lib
from graphsage_amd.ops import project_fwd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_560_000
g = torch.Generator(device="cuda").manual_seed(1)
agg = torch.randn(n, 256, device="cuda", generator=g)
x = torch.randn(n, 256, device="cuda", generator=g)
wl = torch.randn(256, 256, device="cuda", generator=g) / 16
wr = torch.randn(256, 256, device="cuda", generator=g) / 16
b = torch.randn(256, device="cuda", generator=g)
for _ in range(3):
    out, _ = project_fwd(agg, x, wl, b, wr, epilogue=_lib.EPI_RELU)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
out, _ = project_fwd(agg, x, wl, b, wr, epilogue=_lib.EPI_RELU)
e1.record()
torch.cuda.synchronize()
print(f"project_fwd 256->256, N={n}: {e0.elapsed_time(e1):.3f} ms")
lib = _lib.load()
buf2 = np.zeros((2, 256, 8), dtype=np.uint64)
lib.gsage_debug_hb_prof.restype = C.c_int
lib.gsage_debug_hb_prof.argtypes = [C.c_void_p]
assert lib.gsage_debug_hb_prof(buf2.ctypes.data) == 0
names = ["loop top", "split+ds_write", "issue (loads/DMA)", "MFMA x48", "vmcnt wait", "lgkm+barrier", "epilogue", "chunks"]
for wv in range(2):
    buf = buf2[wv]
    ch = buf[:, 7].astype(np.float64)
    print("wave", wv * 4, "(phase order differs between the wave groups)")
    for k in range(7):
        per = buf[:, k].astype(np.float64) / ch
        print(f"  {names[k]:18s} {per.mean():9.1f} cycles/chunk (min {per.min():.1f}, max {per.max():.1f})")
    print("  chunks per workgroup", ch.mean(), " total cycles/chunk", (buf[:, :7].sum(1) / ch).mean())
