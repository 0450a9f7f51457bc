"""GPU: the touched-row hand-over between the loss backward and the layer behind it.
gsage_touched_rows (bit mask + sorted list, integer work: bit-exact vs torch.unique),
gsage_zero_rows, gsage_project_bwd_rows vs the dense gsage_project_bwd_ex, the gather that masks
both operands (dead rows hold NaN on purpose), and the whole backward with and without it."""
import pytest
import torch

from oracle import sage_oracle as O
from util import assert_close

pytestmark = pytest.mark.gpu


def _bits_to_rows(mask, n):
    m = mask.cpu().to(torch.int64) & 0xFFFFFFFF
    bits = ((m.unsqueeze(1) >> torch.arange(32)) & 1).flatten()[:n]
    return bits.nonzero().flatten()


@pytest.mark.parametrize("n,b,k", [(1000, 64, 5), (33, 7, 0), (100_003, 50_000, 5), (5_000_000, 300_000, 5),
                                   (64, 4096, 5)])
def test_touched_rows_bit_exact(n, b, k):
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(n + b)
    t = torch.randint(0, n, (b,), generator=g)
    c = torch.randint(0, n, (b,), generator=g)
    neg = torch.randint(0, n, (b, k), generator=g) if k else torch.zeros(b, 0, dtype=torch.int64)
    if n > 100:
        c[3] = n + 5          # out-of-range ids are ignored (the loss kernel flags them)
        t[5] = -1
    live = ops.touched_rows(t.cuda(), c.cuda(), neg.cuda(), n)
    allidx = torch.cat([t, c, neg.flatten()])
    want = torch.unique(allidx[(allidx >= 0) & (allidx < n)])
    cnt = int(live.n_rows.item())
    assert cnt == want.numel()
    assert torch.equal(live.rows[:cnt].cpu().to(torch.int64), want)       # ascending, exact
    assert torch.equal(_bits_to_rows(live.mask, n), want)
    assert live.cap >= cnt


def test_zero_rows_clears_exactly_the_listed_rows():
    from graphsage_amd import _lib, ops
    n, d = 5000, 64
    t = torch.tensor([5, 17, 17, 4999]).cuda()
    c = torch.tensor([0, 1, 2, 3]).cuda()
    neg = torch.tensor([[100, 200], [300, 400], [500, 600], [700, 800]]).cuda()
    live = ops.touched_rows(t, c, neg, n)
    buf = torch.ones(n, d, device="cuda")
    rc = _lib.load().gsage_zero_rows(live.rows.data_ptr(), live.n_rows.data_ptr(), live.cap, d,
                                     buf.data_ptr(), _lib.stream_handle())
    assert rc == 0
    rows = live.rows[:int(live.n_rows.item())].long()
    assert rows.numel() == 15
    assert float(buf[rows].abs().max()) == 0.0
    keep = torch.ones(n, dtype=torch.bool, device="cuda")
    keep[rows] = False
    assert bool((buf[keep] == 1).all())


@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("n,nlive", [(5000, 777), (70, 70), (100_000, 33), (40_000, 39_999)])
def test_project_bwd_rows_equals_dense_backward(epi, n, nlive):
    """Same grad_w_l / grad_w_r / grad_b, same grad_agg / grad_x on the listed rows; rows outside
    the list are not written."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(n + epi)
    agg, x = torch.randn(n, 64, generator=g).cuda(), torch.randn(n, 64, generator=g).cuda()
    w_l, w_r = (torch.randn(64, 64, generator=g) * 0.2).cuda(), (torch.randn(64, 64, generator=g) * 0.2).cuda()
    b_l = torch.randn(64, generator=g).cuda()
    scale = torch.rand(n, generator=g).cuda() + 0.1
    p = 0.3 if epi == 2 else 0.0
    out, rnorm = ops.project_fwd(agg, x, w_l, b_l, w_r, epi, None, 1234, p)
    rows = torch.randperm(n, generator=g)[:nlive].sort().values
    grad_out = torch.zeros(n, 64)
    grad_out[rows] = torch.randn(nlive, 64, generator=g)
    grad_out = grad_out.cuda()
    dense = ops.project_bwd(grad_out, out, rnorm, agg, x, w_l, w_r, scale, epi, p, True)
    dummy = torch.zeros(1, dtype=torch.int64).cuda()
    live = ops.touched_rows(rows.cuda(), rows[:1].cuda().expand(nlive).contiguous(),
                            torch.zeros(nlive, 0, dtype=torch.int64).cuda(), n)
    assert int(live.n_rows.item()) == nlive
    got = ops.project_bwd_rows(grad_out, out, rnorm, agg, x, w_l, w_r, scale, epi, p, live)
    for name, a, b in zip(("grad_w_l", "grad_b", "grad_w_r"), got[2:], dense[2:]):
        assert_close(a, b, what=name)
    r = rows.cuda()
    assert_close(got[0][r], dense[0][r], what="grad_agg")
    assert_close(got[1][r], dense[1][r], what="grad_x")


def test_gather_live_ignores_whatever_dead_rows_hold():
    from graphsage_amd import default_cache, ops
    from graphsage_amd.synth import powerlaw_graph
    n = 30_000
    ei = powerlaw_graph(n, 14, seed=11, max_degree=4000).cuda()     # has hub rows (> 512 entries)
    graph = default_cache().get(ei, n)
    assert int(graph.n_long[1]) > 0
    g = torch.Generator().manual_seed(1)
    rows = torch.randperm(n, generator=g)[: n // 3].sort().values.cuda()
    live = ops.touched_rows(rows, rows, torch.zeros(rows.numel(), 0, dtype=torch.int64).cuda(), n)
    src = torch.zeros(n, 64, device="cuda")
    add = torch.zeros(n, 64, device="cuda")
    src[rows] = torch.randn(rows.numel(), 64, generator=g).cuda()
    add[rows] = torch.randn(rows.numel(), 64, generator=g).cuda()
    want = ops.aggregate_mean_bwd(graph, src, prescaled=True, add_to=add.clone())
    dead = torch.ones(n, dtype=torch.bool, device="cuda")
    dead[rows] = False
    src[dead] = float("nan")                                        # "may hold anything"
    add[dead] = float("nan")
    got = ops.aggregate_mean_bwd(graph, src, prescaled=True, add_to=add, live_mask=live.mask)
    assert bool(torch.isfinite(got).all())
    assert_close(got, want, rtol=1e-5, floor=2e-6)
    with pytest.raises(ValueError):                                 # the mask is not a hint
        ops.segment_gather(graph.rowptr_t, graph.col_t, torch.zeros(n, 32, device="cuda"),
                           add_to=torch.zeros(n, 32, device="cuda"), live_mask=live.mask)


def test_backward_through_model_uses_the_list_and_matches_dense_and_oracle():
    """loss.backward() through GraphSAGE: the loss backward attaches the row list, the last layer
    consumes it; parameter gradients equal the dense path's and the oracle's."""
    from graphsage_amd import GraphSAGE, SkipGramLoss, ops
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n, b, k = 20_000, 700, 5
    ei = powerlaw_graph(n, 10, seed=42)
    x = memory_features(n, seed=42)
    t, c, neg = cooccurrence_batch(ei, n, b, k, seed=3)
    torch.manual_seed(0)
    ref = O.OracleGraphSAGE(7, 64, 64, 2, 0.0).train()
    m = GraphSAGE(7, 64, 64, 2, 0.0)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().train()
    xs, es, ts, cs, ns = x.cuda(), ei.cuda(), t.cuda(), c.cuda(), neg.cuda()

    def grads(live_on):
        ops.LIVE_ROWS = live_on
        m.zero_grad()
        before = dict(ops.LIVE_STATS)
        loss = SkipGramLoss()(m(xs, es), ts, cs, ns)
        loss.backward()
        used = ops.LIVE_STATS["used"] - before["used"]
        return [p.grad.detach().clone() for p in m.parameters()], used

    try:
        g_live, used = grads(True)
        assert used == 1
        g_dense, used = grads(False)
        assert used == 0
    finally:
        ops.LIVE_ROWS = True
    O.skipgram_loss(ref(x, ei), t, c, neg, 5.0).backward()
    for (name, pr), a, d in zip(ref.named_parameters(), g_live, g_dense):
        assert_close(a, d, what=f"live vs dense {name}")
        assert_close(a, pr.grad, what=f"live vs oracle {name}")
    # a gradient tensor modified in place after the loss backward loses the hand-over
    emb = m(xs, es)
    ge = torch.autograd.grad(SkipGramLoss()(emb, ts, cs, ns), emb, retain_graph=True)[0]
    assert getattr(ge, "_gsage_live", None) is not None
    ge.add_(1.0)                                         # no longer zero outside the batch
    before = ops.LIVE_STATS["used"]
    m.zero_grad()
    emb.backward(ge)
    assert ops.LIVE_STATS["used"] == before


@pytest.mark.parametrize("rowwise", [False, True])
def test_train_step_grad_table_matches_dense_path(rowwise):
    """TrainStep builds d(loss)/d(emb) in a persistent table.  With the atomic loss backward the
    table is cleared row-wise after the backward and is all zero between steps; with the row-wise
    (sorted, deterministic) backward the listed rows are overwritten in full and the layer reads
    through the list, so nothing is cleared and the table is merely marked stale.  Either way the
    parameters after three steps equal the run with the hand-over off (zero-filled gradient,
    dense kernels)."""
    from graphsage_amd import GraphSAGE, TrainStep, ops
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n, b, k = 30_000, 1500, 5
    ei = powerlaw_graph(n, 10, seed=42).cuda()
    x = memory_features(n, seed=42).cuda()
    batches = [cooccurrence_batch(ei, n, b, k, seed=20 + i) for i in range(3)]
    outs = []
    was = ops.LOSS_BWD_ROWS
    try:
        ops.LOSS_BWD_ROWS = rowwise
        for live_on in (True, False):
            ops.LIVE_ROWS = live_on
            torch.manual_seed(0)
            m = GraphSAGE(7, 64, 64, 2, 0.0).cuda()
            ts = TrainStep(m, x, ei)
            for bt in batches:
                ts.step(*bt)
                tab = ts.loss_fn.grad_table
                if live_on:
                    assert tab.buf is not None and not tab.dirty
                    if rowwise:
                        assert tab.stale
                    else:
                        assert float(tab.buf.abs().max()) == 0.0 and not tab.stale
            outs.append([p.detach().clone() for p in m.parameters()])
    finally:
        ops.LIVE_ROWS, ops.LOSS_BWD_ROWS = True, was
    for a, d in zip(*outs):
        assert_close(a, d, rtol=1e-5, floor=2e-6)


def test_stale_gradient_table_is_repaired_when_the_row_list_cannot_be_used():
    """After a row-wise step the table's unlisted rows hold the former step's values.  A backward
    that cannot go through the list (here: the hand-over switched off between loss and layer)
    must see zeros there: SAGELayerFunction rebuilds the dense gradient."""
    from graphsage_amd import GraphSAGE, TrainStep, ops
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n, b, k = 30_000, 1500, 5
    ei = powerlaw_graph(n, 10, seed=42).cuda()
    x = memory_features(n, seed=42).cuda()
    batches = [cooccurrence_batch(ei, n, b, k, seed=40 + i) for i in range(3)]
    finals = []
    try:
        for sabotage in (False, True):
            torch.manual_seed(0)
            m = GraphSAGE(7, 64, 64, 2, 0.0).cuda()
            ts = TrainStep(m, x, ei)
            for i, bt in enumerate(batches):
                ops.SKIP_ZERO_ROWS = not (sabotage and i == 2)   # the layer refuses the list in step 3
                ts.step(*bt)
            finals.append([p.detach().clone() for p in m.parameters()])
    finally:
        ops.SKIP_ZERO_ROWS = True
    for a, d in zip(*finals):
        assert_close(a, d, rtol=1e-5, floor=2e-6)


def test_full_size_c4_backward_with_and_without_the_row_list():
    """BASELINE C4 sizes (10 M nodes, ~200 M edges, 2^20 pairs, K=5), beyond the CPU oracle: the
    whole backward with the touched-row hand-over equals the dense backward (zero-filled gradient,
    every row through the GEMMs, zero-row ballot) parameter gradient by parameter gradient, and
    the row list is exactly the set of rows the batch names."""
    from graphsage_amd import GraphSAGE, SkipGramLoss, ops
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n, b, k = 10_000_000, 1 << 20, 5
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    t, c, neg = cooccurrence_batch(ei, n, b, k, seed=1001)
    torch.manual_seed(0)
    m = GraphSAGE(7, 64, 64, 2, 0.0).cuda().train()
    live = ops.touched_rows(t, c, neg, n)
    cnt = int(live.n_rows.item())
    want = torch.zeros(n, dtype=torch.bool, device="cuda")
    want[t] = True
    want[c] = True
    want[neg.reshape(-1)] = True
    assert cnt == int(want.sum())
    assert torch.equal(live.rows[:cnt].long(), want.nonzero().flatten())
    del want, live
    grads = []
    try:
        for on in (True, False):
            ops.LIVE_ROWS = on
            m.zero_grad()
            used = ops.LIVE_STATS["used"]
            SkipGramLoss(deterministic=False)(m(x, ei), t, c, neg).backward()
            assert ops.LIVE_STATS["used"] - used == (1 if on else 0)
            grads.append([p.grad.detach().clone() for p in m.parameters()])
    finally:
        ops.LIVE_ROWS = True
    for (name, _), a, d in zip(m.named_parameters(), *grads):
        scale = float(d.abs().max())
        assert scale > 0 and bool(torch.isfinite(a).all())
        # float atomics in the loss backward make the last bits run-dependent in both runs
        assert float((a - d).abs().max()) <= 1e-5 * scale, name

@pytest.mark.parametrize("kind", ["independent", "adverse"])
def test_row_list_backward_on_rows_far_apart(kind):
    """The fp16 forms scale dX per gradient row and dW per block of 8 listed rows.  Rows whose gradient and
    operand magnitudes spread independently over 2^+-24 must not cost accuracy, and neither may the adverse
    pattern -- large gradients on the rows with the small operands, alternating inside every block -- while
    the two factors lie 2^16 apart (the limit INTEGRATION.md states).  The reference computes this chain in
    fp32 (autograd of graphsage.py:138); bar: 1e-5 of the largest gradient, against float64."""
    from graphsage_amd import _lib, ops
    n = 60_000
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(17)
    if kind == "independent":
        def mk():
            mag = torch.exp2((torch.rand(n, 1, device=dev, generator=g) * 2 - 1) * 24)
            return torch.randn(n, 64, device=dev, generator=g) * mag
        go, agg, x = mk(), mk(), mk()
        rows = torch.nonzero(torch.rand(n, device=dev, generator=g) < 0.45).flatten().int()
    else:
        alt = (torch.arange(n, device=dev) & 1).float().mul(2).sub(1).unsqueeze(1)
        up, down = torch.exp2(alt * 8), torch.exp2(-alt * 8)
        go = torch.randn(n, 64, device=dev, generator=g) * up
        agg = torch.randn(n, 64, device=dev, generator=g) * down
        x = torch.randn(n, 64, device=dev, generator=g) * down
        rows = torch.arange(n, device=dev, dtype=torch.int32)
    w_l = torch.randn(64, 64, device=dev, generator=g) / 8
    w_r = torch.randn(64, 64, device=dev, generator=g) / 8
    cap = rows.numel()
    live = ops.LiveRows(None, rows, torch.tensor([cap], dtype=torch.int32, device=dev), cap, 0, n)
    ga, gx, gwl, gb, gwr = ops.project_bwd_rows(go, None, None, agg, x, w_l, w_r, torch.ones(n, device=dev),
                                                _lib.EPI_NONE, 0.0, live)
    r = rows.long()
    d = go[r].double()
    for name, got, want in (("grad_w_l", gwl, d.t() @ agg[r].double()), ("grad_b", gb, d.sum(0)),
                            ("grad_w_r", gwr, d.t() @ x[r].double())):
        err = float((got.double() - want).abs().max()) / float(want.abs().max())
        assert err <= 1e-5, (kind, name, err)
    for name, got, w in (("grad_agg", ga, w_l), ("grad_x", gx, w_r)):     # every row against its own largest
        want = d @ w.double()
        rel = (got[r].double() - want).abs().max(1).values / want.abs().max(1).values.clamp_min(1e-300)
        assert float(rel.max()) <= 1e-5, (kind, name, float(rel.max()))


def test_row_list_backward_in_its_other_arithmetic_modes():
    """gsage_project_bwd_rows has three forms: everything on fp32 MFMAs (GSAGE_BWD64=f32), the dX half on
    the fp16 pipe (=dx, bwd64_fused_k<.., HDX>) and, the default, dX and dW both (bwd64h_rows_k).  The
    library reads the switch once, so the two that are not the default run this file's row-list cases
    in a child process each."""
    import os
    import subprocess
    import sys
    if os.environ.get("GSAGE_BWD64"):
        pytest.skip("already inside the forced run")
    here = os.path.dirname(os.path.abspath(__file__))
    for mode in ("f32", "dx"):
        env = dict(os.environ, GSAGE_BWD64=mode)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                            os.path.join(here, "test_gpu_live_rows.py"), "-k",
                            "test_project_bwd_rows_equals_dense_backward or test_backward_through_model"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"GSAGE_BWD64={mode}:\n{r.stdout[-3000:]}\n{r.stderr[-1000:]}"
        assert " passed" in r.stdout
