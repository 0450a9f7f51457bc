"""GPU: K4 / K5 skip-gram loss vs golden vectors produced by the REFERENCE's own loss.py
(tests/golden/loss_golden.npz) and vs the oracle on larger seeded inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import kat
from oracle import sage_oracle as O

from util import assert_close

pytestmark = pytest.mark.gpu
KEYS = ("total_loss", "pos_loss", "neg_loss", "pos_score_mean", "neg_score_mean", "pos_prob", "neg_prob")


def _names(golden_dir):
    z = np.load(os.path.join(golden_dir, "loss_golden.npz"))
    return z, [str(n) for n in z["names"]]


def test_reference_goldens(golden_dir):
    from graphsage_amd import SkipGramLoss
    z, names = _names(golden_dir)
    for name in names:
        emb = torch.from_numpy(z[f"{name}.emb"]).cuda().requires_grad_()
        t, c, n = (torch.from_numpy(z[f"{name}.{k}"]).cuda() for k in ("targets", "contexts", "negatives"))
        fn = SkipGramLoss(scale_factor=float(z[f"{name}.scale"]))
        loss, det = fn.forward_with_details(emb, t, c, n)
        assert loss.dim() == 0
        assert loss.item() == pytest.approx(float(z[f"{name}.loss"]), rel=1e-5), name
        for i, k in enumerate(KEYS):
            assert det[k] == pytest.approx(z[f"{name}.details"][i], rel=1e-5, abs=1e-7), (name, k)
        loss.backward()
        ref = torch.from_numpy(z[f"{name}.grad"])
        assert_close(emb.grad.cpu(), ref, what=str(name))
        assert fn(emb.detach(), t, c, n).item() == pytest.approx(float(z[f"{name}.loss"]), rel=1e-5)


def test_kat5_by_hand():
    from graphsage_amd import SkipGramLoss
    k = kat.KAT5
    loss, det = SkipGramLoss(k["scale"]).forward_with_details(
        torch.tensor(k["emb"]).cuda(), torch.tensor(k["targets"]).cuda(),
        torch.tensor(k["contexts"]).cuda(), torch.tensor(k["negatives"]).cuda())
    assert loss.item() == pytest.approx(k["loss"], rel=1e-6)
    assert det["pos_prob"] == pytest.approx(k["pos_prob"], rel=1e-6)
    assert det["neg_prob"] == pytest.approx(k["neg_prob"], rel=1e-6)


@pytest.mark.parametrize("n,d,b,k", [(100_000, 64, 131072, 5), (5000, 128, 1024, 5), (3000, 256, 513, 9),
                                     (2000, 48, 777, 2), (2000, 6, 100, 5), (1000, 320, 64, 5)])
def test_against_oracle(n, d, b, k):
    from graphsage_amd import SkipGramLoss
    g = torch.Generator().manual_seed(n + d)
    emb = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).requires_grad_()
    t = torch.randint(0, n, (b,), generator=g)
    c = torch.randint(0, n, (b,), generator=g)
    neg = torch.randint(0, n, (b, k), generator=g)
    ref, rdet = O.skipgram_loss_with_details(emb, t, c, neg, 5.0)
    ref.backward()
    e2 = emb.detach().cuda().requires_grad_()
    loss, det = SkipGramLoss(5.0).forward_with_details(e2, t.cuda(), c.cuda(), neg.cuda())
    (loss * 3.0).backward()                                   # upstream gradient != 1
    assert loss.item() == pytest.approx(ref.item(), rel=1e-5)
    for key in KEYS:
        assert det[key] == pytest.approx(rdet[key], rel=1e-5, abs=1e-7)
    assert_close(e2.grad.cpu() / 3.0, emb.grad)


def test_sharded_losses_add_up_to_global_mean():
    from graphsage_amd import SkipGramLoss, shard_bounds
    g = torch.Generator().manual_seed(3)
    n, d, b, k = 4000, 64, 1001, 5
    emb = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).cuda()
    t, c = torch.randint(0, n, (b,), generator=g).cuda(), torch.randint(0, n, (b,), generator=g).cuda()
    neg = torch.randint(0, n, (b, k), generator=g).cuda()
    fn = SkipGramLoss()
    full, stats = fn.forward_stats(emb, t, c, neg)
    parts = torch.zeros(8, device="cuda")
    for r in range(3):
        lo, hi = shard_bounds(b, r, 3)
        _, s = fn.forward_stats(emb, t[lo:hi], c[lo:hi], neg[lo:hi], global_batch=b)
        parts += s
    assert_close(parts[:7], stats[:7])


def test_loss_value_bitwise_reproducible():
    from graphsage_amd import SkipGramLoss
    g = torch.Generator().manual_seed(4)
    emb = torch.randn(10000, 64, generator=g).cuda()
    t, c = torch.randint(0, 10000, (50000,), generator=g).cuda(), torch.randint(0, 10000, (50000,), generator=g).cuda()
    neg = torch.randint(0, 10000, (50000, 5), generator=g).cuda()
    fn = SkipGramLoss()
    _, s1 = fn.forward_stats(emb, t, c, neg)
    _, s2 = fn.forward_stats(emb, t, c, neg)
    assert torch.equal(s1, s2)


def test_out_of_range_pair_index_raises():
    from graphsage_amd import SkipGramLoss
    emb = torch.randn(10, 64).cuda()
    t = torch.tensor([1, 2]).cuda()
    c = torch.tensor([3, 10]).cuda()           # 10 is out of range
    neg = torch.tensor([[1, 2], [3, 4]]).cuda()
    with pytest.raises(IndexError):
        SkipGramLoss().forward_with_details(emb, t, c, neg)


def test_rejects_wrong_inputs():
    from graphsage_amd import SkipGramLoss
    emb = torch.randn(10, 64).cuda()
    t = torch.tensor([1, 2]).cuda()
    neg = torch.tensor([[1, 2], [3, 4]]).cuda()
    with pytest.raises(ValueError):
        SkipGramLoss()(emb, t, t[:1], neg)
    with pytest.raises(ValueError):
        SkipGramLoss()(emb, t.int(), t, neg)
    with pytest.raises(ValueError):
        SkipGramLoss()(emb.cpu(), t.cpu(), t.cpu(), neg.cpu())     # no CPU path


@pytest.mark.parametrize("n,d,b,k,hot", [(5000, 64, 4096, 5, False), (300, 64, 30000, 5, True),
                                         (2000, 128, 1000, 3, False), (100, 30, 500, 7, False),
                                         (50, 64, 7, 0, False)])
def test_deterministic_and_atomic_backward_agree_with_oracle(n, d, b, k, hot):
    """K5 in both forms.  `hot`: every pair shares one target, so its row of the per-batch CSR
    (b*(1+k) terms) goes through the hub-row path."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(n + b)
    emb = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).requires_grad_()
    t = torch.full((b,), 5, dtype=torch.int64) if hot else torch.randint(0, n, (b,), generator=g)
    c = torch.randint(0, n, (b,), generator=g)
    neg = torch.randint(0, n, (b, k), generator=g)
    if k > 0:
        O.skipgram_loss(emb, t, c, neg, 5.0).backward()
    else:   # no negatives: only the positive term exists
        (-torch.nn.functional.logsigmoid((emb[t] * emb[c]).sum(1) * 5.0).mean()).backward()
    e, tt, cc, nn_ = emb.detach().cuda(), t.cuda(), c.cuda(), neg.cuda()
    coef, _ = ops.skipgram_fwd(e, tt, cc, nn_, 5.0)
    one = torch.ones((), device="cuda")
    det1 = ops.skipgram_bwd(e, tt, cc, nn_, coef, one, deterministic=True)
    det2 = ops.skipgram_bwd(e, tt, cc, nn_, coef, one, deterministic=True)
    atom = ops.skipgram_bwd(e, tt, cc, nn_, coef, one, deterministic=False)
    assert torch.equal(det1, det2)                              # bitwise reproducible
    assert_close(det1, emb.grad, rtol=1e-5, floor=2e-6, what="deterministic")
    # float atomics add in arrival order: with every pair on one target row that row is a sum of
    # b*(1+k) terms in a different order each run, so its bar is wider
    assert_close(atom, emb.grad, rtol=1e-5, floor=2e-5 if hot else 2e-6, what="atomic")


def test_deterministic_backward_ignores_out_of_range_pairs():
    from graphsage_amd import ops
    emb = torch.randn(10, 64).cuda()
    t = torch.tensor([1, 2, 3]).cuda()
    c = torch.tensor([3, 10, 4]).cuda()                         # pair 1 has a bad context id
    neg = torch.tensor([[1, 2], [3, 4], [5, -1]]).cuda()        # pair 2 has a bad negative
    coef, stats = ops.skipgram_fwd(emb, t, c, neg, 5.0)
    assert stats[7].item() != 0
    one = torch.ones((), device="cuda")
    d = ops.skipgram_bwd(emb, t, c, neg, coef, one, deterministic=True)
    a = ops.skipgram_bwd(emb, t, c, neg, coef, one, deterministic=False)
    assert torch.isfinite(d).all()
    assert_close(d, a, rtol=1e-5, floor=2e-6)


def test_full_size_c4_loss_properties():
    """BASELINE C4 sizes (10 M x 64 table, 2^20 pairs, K=5), where the oracle cannot run:
    the two independent backward algorithms agree row for row, the column sums of grad_emb equal
    the analytic ones (float64 index arithmetic in torch), the loss of the batch is the
    shard-weighted sum of 8 shard losses, and untouched rows get exactly zero."""
    from graphsage_amd import ops
    from graphsage_amd.synth import cooccurrence_batch, powerlaw_graph
    n, d, b, k = 10_000_000, 64, 1 << 20, 5
    g = torch.Generator(device="cuda").manual_seed(3)
    emb = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
    ei = powerlaw_graph(1_000_000, 20, seed=5, device="cuda")          # index source only
    t, c, neg = cooccurrence_batch(ei, 1_000_000, b, k, seed=9)
    del ei
    spread = torch.randint(0, 10, (b,), device="cuda", generator=g)    # spread ids over all 10 M rows
    t, c, neg = t * 10 + spread, c * 10 + spread, neg * 10 + spread.view(-1, 1)
    coef, stats = ops.skipgram_fwd(emb, t, c, neg, 5.0)
    assert stats[7].item() == 0 and torch.isfinite(stats[:7]).all()
    # loss = sum of shard losses normalised by the global batch
    parts = []
    for s in range(8):
        sl = slice(s * b // 8, (s + 1) * b // 8)
        _, st = ops.skipgram_fwd(emb, t[sl], c[sl], neg[sl], 5.0, norm_b=b, norm_bk=b * k)
        parts.append(st[0])
    assert abs(float(torch.stack(parts).sum()) - float(stats[0])) <= 2e-6 * abs(float(stats[0]))
    one = torch.ones((), device="cuda")
    atom = ops.skipgram_bwd(emb, t, c, neg, coef, one, deterministic=False)
    det = ops.skipgram_bwd(emb, t, c, neg, coef, one, deterministic=True)
    scale = float(det.abs().max())
    assert float((atom - det).abs().max()) <= 1e-5 * scale
    touched = torch.zeros(n, dtype=torch.bool, device="cuda")
    touched[t] = True
    touched[c] = True
    touched[neg.reshape(-1)] = True
    assert float(det[~touched].abs().max()) == 0.0 and float(atom[~touched].abs().max()) == 0.0
    # analytic column sums: every term coef*emb[src] lands once in grad_emb
    cf = coef.double()
    others = torch.cat([c.view(-1, 1), neg], 1)                          # [B, 1+K]
    want = (cf.unsqueeze(2) * emb[others].double()).sum((0, 1))          # terms added to targets
    want += (cf.unsqueeze(2) * emb[t].double().unsqueeze(1)).sum((0, 1)) # terms added to c / negatives
    got = det.double().sum(0)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9


@pytest.mark.parametrize("n,d,b,k,hot", [(50_000, 64, 4096, 5, False), (3000, 64, 3000, 5, True),
                                         (20_000, 128, 1000, 3, False), (700, 64, 50, 7, False),
                                         (64, 64, 7, 0, False), (33, 64, 1, 5, False)])
def test_rowwise_backward_over_the_touched_rows(n, d, b, k, hot):
    """gsage_skipgram_bwd_rows: the sorted sums over the LISTED rows only.  Listed rows equal the
    oracle's gradient (autograd through loss.py:97-136's op sequence) and the full sorted form;
    every other row of the buffer is left exactly as it was (NaN here); two runs agree bit for
    bit.  `hot`: one target shared by every pair (a listed row with b*(1+k) terms)."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(n + b)
    emb = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).requires_grad_()
    t = torch.full((b,), 5, dtype=torch.int64) if hot else torch.randint(0, n, (b,), generator=g)
    c = torch.randint(0, n, (b,), generator=g)
    neg = torch.randint(0, n, (b, k), generator=g)
    if k > 0:
        O.skipgram_loss(emb, t, c, neg, 5.0).backward()
    else:
        (-torch.nn.functional.logsigmoid((emb[t] * emb[c]).sum(1) * 5.0).mean()).backward()
    e, tt, cc, nn_ = emb.detach().cuda(), t.cuda(), c.cuda(), neg.cuda()
    coef, _ = ops.skipgram_fwd(e, tt, cc, nn_, 5.0)
    one = torch.ones((), device="cuda")
    live = ops.touched_rows(tt, cc, nn_, n)
    cnt = int(live.n_rows.item())
    rows = live.rows[:cnt].long()
    # the word prefix is the rank of each word's first listed row
    words = (n + 31) // 32
    bits = torch.zeros(words * 32, dtype=torch.int64, device="cuda")
    bits[rows] = 1
    want_pre = torch.cumsum(bits.view(words, 32).sum(1), 0) - bits.view(words, 32).sum(1)
    assert torch.equal(live.word_prefix.long(), want_pre)
    outs = []
    for _ in range(2):
        buf = torch.full((n, d), float("nan"), device="cuda")
        ops.skipgram_bwd_rows(e, tt, cc, nn_, coef, one, buf, live)
        outs.append(buf)
    assert torch.equal(outs[0][rows], outs[1][rows])                    # bitwise reproducible
    untouched = torch.ones(n, dtype=torch.bool, device="cuda")
    untouched[rows] = False
    assert bool(torch.isnan(outs[0][untouched]).all())                  # not written at all
    got = torch.zeros(n, d, device="cuda")
    got[rows] = outs[0][rows]
    assert_close(got, emb.grad, rtol=1e-5, floor=2e-6, what="row-wise vs oracle")
    det = ops.skipgram_bwd(e, tt, cc, nn_, coef, one, deterministic=True)
    assert_close(got, det, rtol=1e-5, floor=2e-6, what="row-wise vs sorted")


def test_rowwise_backward_ignores_out_of_range_pairs():
    from graphsage_amd import ops
    emb = torch.randn(40, 64).cuda()
    t = torch.tensor([1, 2, 3]).cuda()
    c = torch.tensor([3, 40, 4]).cuda()                         # pair 1 has a bad context id
    neg = torch.tensor([[1, 2], [3, 4], [5, -1]]).cuda()        # pair 2 has a bad negative
    coef, stats = ops.skipgram_fwd(emb, t, c, neg, 5.0)
    assert stats[7].item() != 0
    one = torch.ones((), device="cuda")
    live = ops.touched_rows(t, c, neg, 40)
    buf = torch.zeros(40, 64, device="cuda")
    ops.skipgram_bwd_rows(emb, t, c, neg, coef, one, buf, live)
    a = ops.skipgram_bwd(emb, t, c, neg, coef, one, deterministic=False)
    assert torch.isfinite(buf).all()
    assert_close(buf, a, rtol=1e-5, floor=2e-6)


def test_full_size_c4_rowwise_backward():
    """C4 sizes (10 M x 64 table, 2^20 pairs, K = 5): the row-wise sums against the full sorted
    form and the atomics, untouched rows untouched, analytic column sums, bitwise repeatable."""
    from graphsage_amd import ops
    from graphsage_amd.synth import cooccurrence_batch, powerlaw_graph
    n, d, b, k = 10_000_000, 64, 1 << 20, 5
    g = torch.Generator(device="cuda").manual_seed(3)
    emb = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    t, c, neg = cooccurrence_batch(ei, n, b, k, seed=1001)              # the benchmark's batches
    del ei
    coef, _ = ops.skipgram_fwd(emb, t, c, neg, 5.0)
    one = torch.ones((), device="cuda")
    live = ops.touched_rows(t, c, neg, n)
    cnt = int(live.n_rows.item())
    rows = live.rows[:cnt].long()
    a = torch.zeros(n, d, device="cuda")
    ops.skipgram_bwd_rows(emb, t, c, neg, coef, one, a, live)
    b2 = torch.full((n, d), 7.0, device="cuda")
    ops.skipgram_bwd_rows(emb, t, c, neg, coef, one, b2, live)
    assert torch.equal(a[rows], b2[rows])
    b2[rows] = 7.0
    assert bool((b2 == 7.0).all())                                      # nothing else written
    del b2
    det = ops.skipgram_bwd(emb, t, c, neg, coef, one, deterministic=True)
    scale = float(det.abs().max())
    assert float((a - det).abs().max()) <= 2e-6 * scale
    atom = ops.skipgram_bwd(emb, t, c, neg, coef, one, deterministic=False)
    assert float((a - atom).abs().max()) <= 1e-5 * scale
