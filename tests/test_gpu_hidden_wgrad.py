"""GPU: K7h, the input layer's weight gradients taken from the next layer's backward
(gsage_hidden1_wgrad), against the chain it replaces written out in fp64: scatter of the
aggregation's gradient to the neighbours, ReLU'/dropout' of the stored activation, dW = dpre^T [agg | x]."""
import pytest
import torch

from graphsage_amd import _lib, ops
from graphsage_amd.graph import build_csr

pytestmark = pytest.mark.gpu


def _graph(n, m, hub_deg, seed):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (m,), generator=g)
    dst = torch.randint(0, n, (m,), generator=g)
    if hub_deg:                                  # one destination with a very long row, one isolated node
        src = torch.cat([src, torch.randint(0, n, (hub_deg,), generator=g)])
        dst = torch.cat([dst, torch.full((hub_deg,), 5)])
    keep = (src != n - 1) & (dst != n - 1)
    return torch.stack([src[keep], dst[keep]])


def _case(n, m, hub_deg, frac_live, seed, p=0.3):
    ei = _graph(n, m, hub_deg, seed).cuda()
    graph = build_csr(ei, n)
    g = torch.Generator().manual_seed(seed + 1)
    x8 = torch.randn(n, 8, generator=g)
    x8[:, 7] = 0
    x8 = x8.cuda()
    wl, wr, bl = [t.cuda() for t in (torch.randn(64, 8, generator=g) / 3, torch.randn(64, 8, generator=g) / 3,
                                     torch.randn(64, generator=g) / 3)]
    agg1 = ops.aggregate_mean(graph, x8)
    box = {}
    h1, _ = ops.project_fwd(agg1, x8, wl, bl, wr, epilogue=_lib.EPI_RELU_DROPOUT, seed=1234 + seed, p=p, u_out=box)
    assert box["u"] is not None
    live_ids = torch.nonzero(torch.rand(n, generator=g) < frac_live).flatten()
    if hub_deg:
        live_ids = torch.unique(torch.cat([live_ids, torch.tensor([5, n - 1])]))     # the hub row, the isolated one
    cap = max(1, min(n, live_ids.numel() + 7))
    rows = torch.zeros(cap, dtype=torch.int32)
    rows[:live_ids.numel()] = live_ids.int()
    live = ops.LiveRows(None, rows.cuda(), torch.tensor([live_ids.numel()], dtype=torch.int32).cuda(), cap, 0, n)
    ga = torch.full((n, 64), float("nan"))
    gx = torch.full((n, 64), float("nan"))                   # rows outside the list must not be read
    scale = 10.0 ** torch.randint(-4, 2, (live_ids.numel(), 1), generator=g).float()
    ga[live_ids] = torch.randn(live_ids.numel(), 64, generator=g) * scale
    gx[live_ids] = torch.randn(live_ids.numel(), 64, generator=g) * scale
    return graph, ei, x8, agg1, h1, box["u"], live, live_ids, ga, gx, p


def _expected(ei, n, x8, agg1, h1, live_ids, ga, gx, p):
    src, dst = ei[0].cpu(), ei[1].cpu()
    is_live = torch.zeros(n, dtype=torch.bool)
    is_live[live_ids] = True
    e = is_live[dst]
    gh = torch.zeros(n, 64, dtype=torch.float64)
    gh.index_add_(0, src[e], ga.double()[dst[e]])
    gh[live_ids] += gx.double()[live_ids]
    dpre = gh * (h1.cpu() > 0) / (1.0 - p)
    return dpre.t() @ agg1.cpu().double(), dpre.sum(0), dpre.t() @ x8.cpu().double()


def _gmax(ga, gx, live_ids):
    m = max(ga[live_ids].abs().max().item(), gx[live_ids].abs().max().item()) if live_ids.numel() else 0.0
    return torch.tensor([m], dtype=torch.float32).view(torch.int32).cuda()


@pytest.mark.parametrize("n,m,hub,frac", [(3000, 40000, 2500, 0.5), (500, 3000, 0, 1.0), (70, 200, 0, 0.1),
                                          (20000, 400000, 9000, 0.4)])
def test_input_layer_weight_gradients_from_the_next_layers_backward(n, m, hub, frac):
    graph, ei, x8, agg1, h1, u, live, live_ids, ga, gx, p = _case(n, m, hub, frac, seed=n)
    gwl, gb, gwr = ops.hidden1_wgrad(graph, u, ga.cuda(), gx.cuda(), _gmax(ga, gx, live_ids), live,
                                     _lib.EPI_RELU_DROPOUT, p)
    ewl, eb, ewr = _expected(ei, n, x8, agg1, h1, live_ids, ga, gx, p)
    for got, want in ((gwl, ewl), (gb, eb), (gwr, ewr)):
        tol = 1e-5 * want.abs().max().item()
        assert (got.cpu().double() - want).abs().max().item() <= tol
    assert torch.count_nonzero(gwl[:, 7]).item() == 0 and torch.count_nonzero(gwr[:, 7]).item() == 0
    # a fixed order of summation: the same bits again
    again = ops.hidden1_wgrad(graph, u, ga.cuda(), gx.cuda(), _gmax(ga, gx, live_ids), live,
                              _lib.EPI_RELU_DROPOUT, p)
    assert all(torch.equal(a, b) for a, b in zip(again, (gwl, gb, gwr)))


def test_empty_row_list_gives_zero_gradients():
    graph, ei, x8, agg1, h1, u, live, live_ids, ga, gx, p = _case(300, 2000, 0, 0.5, seed=3)
    live.n_rows.zero_()
    gwl, gb, gwr = ops.hidden1_wgrad(graph, u, ga.cuda(), gx.cuda(), _gmax(ga, gx, live_ids), live,
                                     _lib.EPI_RELU_DROPOUT, p)
    assert not gwl.any() and not gb.any() and not gwr.any()


@pytest.mark.parametrize("same_sign", [False, True])
def test_full_size_c4_against_float64(same_sign):
    """10 M nodes / 200 M edges, the rows a 2^20-pair batch touches (134 M live edges): float64 on the
    GPU, edges in chunks.  Same-sign gradients are the worst case for the matrix pipe's truncating
    accumulate (1.5e-5 before the chains were cut every 8 groups, 8e-7 after); random signs cancel
    heavily (sums of 1e8 terms), so the bar there is the 1e-5 of the weight gradients' parity bar."""
    from graphsage_amd import default_cache
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    nodes, p = 10_000_000, 0.3
    dev = torch.device("cuda:0")
    ei = powerlaw_graph(nodes, 20, seed=42, device=dev)
    graph = default_cache().get(ei, nodes)
    x8 = default_cache().padded(memory_features(nodes, seed=42, device=dev), 8, keep=False)
    g = torch.Generator(device=dev).manual_seed(5)
    wl, wr = torch.randn(64, 8, device=dev, generator=g) / 3, torch.randn(64, 8, device=dev, generator=g) / 3
    bl = torch.randn(64, device=dev, generator=g) / 3
    agg1 = ops.aggregate_mean(graph, x8)
    box = {}
    h1, _ = ops.project_fwd(agg1, x8, wl, bl, wr, epilogue=_lib.EPI_RELU_DROPOUT, seed=77, p=p, u_out=box)
    live = ops.touched_rows(*cooccurrence_batch(ei, nodes, 1 << 20, 5, seed=1001), nodes)
    rows = live.rows[:int(live.n_rows.item())].long()
    is_live = torch.zeros(nodes, dtype=torch.bool, device=dev)
    is_live[rows] = True
    sel = is_live[ei[1]]
    src, dst = ei[0][sel], ei[1][sel]
    del sel, is_live
    ga = torch.zeros(nodes, 64, device=dev)
    gx = torch.zeros(nodes, 64, device=dev)
    r1 = torch.randn(rows.numel(), 64, device=dev, generator=g)
    r2 = torch.randn(rows.numel(), 64, device=dev, generator=g)
    if same_sign:
        r1, r2 = r1.abs(), r2.abs()
    ga[rows] = r1 * graph.inv_deg[rows, None]          # prescaled, as gsage_project_bwd_rows leaves it
    gx[rows] = r2
    del r1, r2
    gmax = torch.maximum(ga.abs().max(), gx.abs().max()).reshape(1).view(torch.int32)
    gwl, gb, gwr = ops.hidden1_wgrad(graph, box["u"], ga, gx, gmax, live, _lib.EPI_RELU_DROPOUT, p)
    gh = torch.zeros(nodes, 64, dtype=torch.float64, device=dev)
    for lo in range(0, src.numel(), 10_000_000):
        gh.index_add_(0, src[lo:lo + 10_000_000], ga[dst[lo:lo + 10_000_000]].double())
    gh[rows] += gx[rows].double()
    gh *= (h1 > 0)
    gh /= (1.0 - p)
    bar = 2e-6 if same_sign else 1e-5
    for got, want in ((gwl, gh.t() @ agg1.double()), (gb, gh.sum(0)), (gwr, gh.t() @ x8.double())):
        assert (got.double() - want).abs().max().item() <= bar * want.abs().max().item()


@pytest.mark.parametrize("n,b", [(60_000, 3000), (10_000_000, 1 << 20)])
def test_model_backward_with_and_without_the_fused_weight_gradients(n, b, monkeypatch):
    """loss.backward() through GraphSAGE(dropout 0.3).train(): with the hand-over on, layer 2's
    backward deposits (grad_agg, grad_x, rows) and layer 1 answers from gsage_hidden1_wgrad; off,
    the masked transposed gather + the first layer's dW kernel run.  Same masks (the dropout stream
    is seeded from (base, step, layer)), same deterministic loss backward: all six parameter
    gradients agree to the 1e-5 bar, at a size the CPU oracle covers and at BASELINE C4."""
    from graphsage_amd import GraphSAGE, SkipGramLoss
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    monkeypatch.setattr(ops, "HIDDEN1_MIN_BYTES", 0)       # the recompute gather (and its row table) at any size
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    t, c, neg = cooccurrence_batch(ei, n, b, 5, seed=1001)
    torch.manual_seed(0)
    m = GraphSAGE(7, 64, 64, 2, 0.3).cuda().train()
    grads = []
    for on in (True, False):
        monkeypatch.setattr(ops, "HIDDEN1_WGRAD", on)
        m.set_dropout_stream(1234, 7)
        m.zero_grad()
        used = ops.WGRAD_STATS["used"]
        SkipGramLoss()(m(x, ei), t, c, neg).backward()
        assert ops.WGRAD_STATS["used"] - used == (1 if on else 0)
        grads.append([p.grad.detach().clone() for p in m.parameters()])
    for (name, _), a, d in zip(m.named_parameters(), *grads):
        scale = float(d.abs().max())
        assert scale > 0 and bool(torch.isfinite(a).all())
        assert float((a - d).abs().max()) <= 1e-5 * scale, name
    # the layers behind the input layer took the same path in both runs: bit-identical there
    assert torch.equal(grads[0][3], grads[1][3]) and torch.equal(grads[0][5], grads[1][5])


def test_captured_step_with_the_fused_weight_gradients(monkeypatch):
    """The whole step as one HIP graph with the recompute gather and gsage_hidden1_wgrad inside
    (work counters and range words are zeroed by kernels in the capture): four replays train like
    four replays of a stepper captured with the hand-over off (same base seed, so the same masks
    from the steppers' Philox generators)."""
    from graphsage_amd import GraphSAGE, TrainStep
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    from util import assert_close
    monkeypatch.setattr(ops, "HIDDEN1_MIN_BYTES", 0)
    n, bs = 30_000, 2048
    ei = powerlaw_graph(n, 12, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    torch.manual_seed(0)
    models = [GraphSAGE(7, 64, 64, 2, 0.3).cuda() for _ in range(2)]
    models[1].load_state_dict(models[0].state_dict())
    steppers = []
    for m, on in zip(models, (True, False)):
        monkeypatch.setattr(ops, "HIDDEN1_WGRAD", on)
        ts = TrainStep(m, x, ei, seed=99)
        used = ops.WGRAD_STATS["used"]
        ts.capture_step(batch_size=bs, num_negatives=5, warmup=1, preserve_state=True)
        assert (ops.WGRAD_STATS["used"] > used) == on
        steppers.append(ts)
    for s in range(4):
        b = cooccurrence_batch(ei, n, bs, 5, seed=20 + s)
        stats = [ts.step_captured(*b).clone() for ts in steppers]
        assert_close(stats[0][:7], stats[1][:7], rtol=2e-5, floor=1e-6, what=f"stats step {s}")
    for (name, pa), (_, pb) in zip(models[0].named_parameters(), models[1].named_parameters()):
        assert_close(pa, pb, rtol=1e-4, floor=2e-5, what=name)


def test_training_backward_through_the_fused_paths_against_the_cpu_oracle(monkeypatch):
    """The oracle's step (PyG's op sequence + loss.py on the CPU, autograd) against the model with
    every round-3 path forced on at a size the oracle finishes in seconds: the recompute gather on
    the rows the input layer writes, the 64 -> 64 forward on the fp16 pipe, the row-list backward
    with its fp16 dX, and the input layer's weight gradients from gsage_hidden1_wgrad -- injected
    dropout masks, all six parameter gradients."""
    from oracle import sage_oracle as O
    from util import assert_close
    from graphsage_amd import GraphSAGE, SkipGramLoss
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    monkeypatch.setattr(ops, "HIDDEN1_MIN_BYTES", 0)
    n, b, k, p = 20_000, 1500, 5, 0.3
    ei = powerlaw_graph(n, 10, seed=42)
    x = memory_features(n, seed=42)
    t, c, neg = cooccurrence_batch(ei, n, b, k, seed=3)
    torch.manual_seed(0)
    ref = O.OracleGraphSAGE(7, 64, 64, 2, p).train()
    m = GraphSAGE(7, 64, 64, 2, p)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().train()
    g = torch.Generator().manual_seed(9)
    keep = (torch.rand(n, 64, generator=g) >= p).to(torch.uint8)
    used_w, used_h = ops.WGRAD_STATS["used"], ops.HIDDEN1_STATS["used"]
    loss = SkipGramLoss()(m(x.cuda(), ei.cuda(), keep_masks=[keep.cuda()]), t.cuda(), c.cuda(), neg.cuda())
    loss.backward()
    assert ops.WGRAD_STATS["used"] == used_w + 1 and ops.HIDDEN1_STATS["used"] == used_h + 1
    want = O.skipgram_loss(ref(x, ei, keep_masks=[keep.float()]), t, c, neg, 5.0)
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
    for (name, pg), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        assert_close(pg.grad.cpu(), pr.grad, rtol=2e-5, floor=1e-5, what=name)
