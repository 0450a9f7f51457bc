"""Timing of the three loss-backward forms at C4 size (10 M x 64 table, 2^20 pairs, K = 5):
atomic scatter (+ the row-wise clear it needs), full sorted form, sorted over the touched rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphsage_amd import ops
from graphsage_amd.synth import cooccurrence_batch, powerlaw_graph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
k, d = 5, 64
ei = powerlaw_graph(n, 20, seed=42, device="cuda")
t, c, neg = cooccurrence_batch(ei, n, b, k, seed=1001)
del ei
emb = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1)
coef, _ = ops.skipgram_fwd(emb, t, c, neg, 5.0)
one = torch.ones((), device="cuda")
buf = torch.zeros(n, d, device="cuda")


def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


live = ops.touched_rows(t, c, neg, n)
print("rows listed", int(live.n_rows.item()), "terms", 2 * (1 + k) * b)
print("touched_rows        %.3f ms" % timed(lambda: ops.touched_rows(t, c, neg, n)))
print("atomic              %.3f ms" % timed(lambda: ops.skipgram_bwd(emb, t, c, neg, coef, one, grad_emb=buf, deterministic=False)))
print("sorted, all N rows  %.3f ms" % timed(lambda: ops.skipgram_bwd(emb, t, c, neg, coef, one, deterministic=True), 5))
print("sorted, listed rows %.3f ms" % timed(lambda: ops.skipgram_bwd_rows(emb, t, c, neg, coef, one, buf, live)))
