"""GPU: the drop-in GraphSAGE module and the training step vs the oracle, with the
invariants the reference's own tests assert (tests/test_model.py:45-99, 126-156)."""
import os

import numpy as np
import pytest
import torch

from oracle import sage_oracle as O

from util import assert_close

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 1e-6


def _pair(cfg, seed=0):
    """(product model on GPU, oracle model on CPU) with identical weights."""
    from graphsage_amd import GraphSAGE
    torch.manual_seed(seed)
    ref = O.OracleGraphSAGE(**cfg)
    m = GraphSAGE(**cfg)
    m.load_state_dict(ref.state_dict(), strict=True)      # PyG-compatible keys both sides
    return m.cuda(), ref


CFGS = [
    dict(in_channels=7, hidden_channels=64, out_channels=64, num_layers=2, dropout=0.3),   # default
    dict(in_channels=7, hidden_channels=128, out_channels=128, num_layers=2, dropout=0.3),  # C3
    dict(in_channels=7, hidden_channels=256, out_channels=256, num_layers=3, dropout=0.0),  # C5
    dict(in_channels=7, hidden_channels=32, out_channels=32, num_layers=2, dropout=0.3),   # ref integration test
    dict(in_channels=7, hidden_channels=64, out_channels=48, num_layers=1, dropout=0.3),   # num_layers==1 quirk
    dict(in_channels=7, hidden_channels=64, out_channels=64, num_layers=2, dropout=0.3, normalize_output=False),
]


@pytest.mark.parametrize("cfg", CFGS)
def test_eval_forward_matches_oracle(cfg):
    m, ref = _pair(cfg)
    n = 700
    x = torch.randn(n, 7)
    ei = torch.randint(0, n, (2, 5000))
    m.eval(), ref.eval()
    with torch.no_grad():
        got = m(x.cuda(), ei.cuda())
        want = ref(x, ei)
    assert got.shape == want.shape
    assert_close(got.cpu(), want)


@pytest.mark.parametrize("cfg", CFGS[:4])
def test_train_forward_backward_with_injected_masks(cfg):
    m, ref = _pair(cfg, seed=1)
    n = 600
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 7, generator=g)
    ei = torch.randint(0, n, (2, 6000), generator=g)
    p = cfg["dropout"]
    nh = cfg["num_layers"] - 1
    keep = [(torch.rand(n, cfg["hidden_channels"], generator=g) >= p).to(torch.uint8) for _ in range(nh)]
    gout = torch.randn(n, cfg["out_channels"], generator=g)
    m.train(), ref.train()
    want = ref(x, ei, keep_masks=[k.float() for k in keep])
    want.backward(gout)
    got = m(x.cuda(), ei.cuda(), keep_masks=[k.cuda() for k in keep])
    got.backward(gout.cuda())
    assert_close(got.detach().cpu(), want.detach())
    for (name, pg), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        assert pg.grad is not None, name
        assert_close(pg.grad.cpu(), pr.grad, rtol=2e-5, floor=1e-5, what=str(name))


def test_input_gradient_when_features_require_grad():
    cfg = CFGS[0]
    m, ref = _pair(cfg, seed=2)
    m.eval(), ref.eval()
    n = 300
    x = torch.randn(n, 7)
    ei = torch.randint(0, n, (2, 2500))
    xr = x.clone().requires_grad_()
    ref(xr, ei).pow(2).sum().backward()      # sum of squared unit rows is constant -> use a weight
    xr.grad = None
    w = torch.randn(n, 64)
    (ref(xr, ei) * w).sum().backward()
    xg = x.cuda().requires_grad_()
    (m(xg, ei.cuda()) * w.cuda()).sum().backward()
    assert_close(xg.grad.cpu(), xr.grad)


def test_reference_test_invariants():
    """tests/test_model.py:23-99 of the reference, on the drop-in class."""
    from graphsage_amd import GraphSAGE
    torch.manual_seed(0)
    model = GraphSAGE(in_channels=7, hidden_channels=64, out_channels=64, num_layers=2, dropout=0.3).cuda()
    features = torch.randn(100, 7).cuda()
    edge_index = torch.randint(0, 100, (2, 500)).cuda()
    model.eval()
    out = model(features, edge_index)
    assert out.shape == (100, 64)
    assert torch.allclose(out.norm(p=2, dim=1), torch.ones(100, device="cuda"), atol=1e-5)
    model.train()
    model(features, edge_index).sum().backward()
    for name, p in model.named_parameters():
        assert p.grad is not None, name
    assert 0 < model.count_parameters() < 100000 and model.count_parameters() == 9216
    assert model.get_embedding_dim() == 64 and model.in_channels == 7
    with pytest.raises(ValueError):
        GraphSAGE(aggregator="lstm")


def test_train_mode_dropout_is_seeded_and_scaled():
    from graphsage_amd import GraphSAGE
    torch.manual_seed(3)
    model = GraphSAGE(dropout=0.5, normalize_output=False).cuda().train()
    x = torch.randn(2000, 7).cuda()
    ei = torch.randint(0, 2000, (2, 20000)).cuda()
    torch.manual_seed(7)
    a = model(x, ei)
    torch.manual_seed(7)
    b = model(x, ei)
    c = model(x, ei)
    assert torch.equal(a, b) and not torch.equal(a, c)        # reproducible under manual_seed
    model.eval()
    e = model(x, ei)
    assert not torch.allclose(a, e)


def test_c1_fixture_embeddings_match_snapshot(golden_dir):
    from graphsage_amd import GraphSAGE
    z = np.load(os.path.join(golden_dir, "c1_graph.npz"))
    s = np.load(os.path.join(golden_dir, "oracle_c1_snapshot.npz"))
    torch.manual_seed(0)
    ref = O.OracleGraphSAGE(7, 64, 64, 2, 0.3)
    m = GraphSAGE(7, 64, 64, 2, 0.3)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().eval()
    emb = m(torch.from_numpy(z["features"]).cuda(), torch.from_numpy(z["edge_index"]).cuda())
    assert_close(emb.detach().cpu(), torch.from_numpy(s["embeddings"]))


def test_training_steps_track_the_oracle():
    """Five full steps (trainer.py:213-240): zero_grad, fwd, loss, bwd, clip, Adam."""
    from graphsage_amd import GraphSAGE, TrainStep
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n, b, k = 3000, 1024, 5
    ei = powerlaw_graph(n, 10, seed=42)
    x = memory_features(n, seed=42)
    cfg = dict(in_channels=7, hidden_channels=64, out_channels=64, num_layers=2, dropout=0.0)
    m, ref = _pair(cfg, seed=4)
    ts = TrainStep(m, x.cuda(), ei.cuda())
    opt = O.make_adam(ref)
    ref.train()
    for step in range(5):
        t, c, neg = cooccurrence_batch(ei, n, b, k, seed=100 + step)
        rl, rdet = O.train_step(ref, opt, x, ei, t, c, neg)
        det = ts.step_with_details(t.cuda(), c.cuda(), neg.cuda())
        assert det["total_loss"] == pytest.approx(rdet["total_loss"], rel=2e-5), step
        assert det["pos_prob"] == pytest.approx(rdet["pos_prob"], rel=2e-5), step
    for (name, pg), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        assert_close(pg.detach().cpu(), pr.detach(), rtol=1e-4, floor=2e-5, what=str(name))
    emb = ts.get_embeddings()
    ref.eval()
    assert_close(emb.cpu(), ref(x, ei).detach(), rtol=1e-4, floor=2e-5)


def test_whole_step_bitwise_reproducible_with_deterministic_loss():
    """With the sort-based loss backward nothing in the step uses atomics: two runs from the
    same state give bit-identical parameters (the survey's cheap race detector)."""
    from graphsage_amd import GraphSAGE, SkipGramLoss, TrainStep
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n = 20000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    batches = [cooccurrence_batch(ei, n, 8192, 5, seed=s) for s in range(3)]
    finals = []
    for run in range(2):
        torch.manual_seed(0)
        m = GraphSAGE(7, 64, 64, 2, 0.3).cuda()
        ts = TrainStep(m, x, ei, loss_fn=SkipGramLoss(deterministic=True))
        torch.manual_seed(123)                                   # same dropout seeds
        for b in batches:
            ts.step(*b)
        finals.append(torch.cat([p.detach().flatten() for p in m.parameters()]).clone())
    assert torch.equal(finals[0], finals[1])


def test_default_step_is_bitwise_reproducible_where_the_row_list_exists():
    """The configuration class of the benchmark's headline (batch small against the table, so the
    loss backward runs row-wise over the touched rows; default SkipGramLoss, dropout 0.3 from the
    in-kernel hash, persistent gradient table left stale between steps): no atomics anywhere, two
    runs from the same state end in bit-identical parameters."""
    from graphsage_amd import GraphSAGE, TrainStep, ops
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n, b, k = 200_000, 16384, 5
    assert ops.loss_bwd_rowwise(n, 64, b, k)
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    batches = [cooccurrence_batch(ei, n, b, k, seed=s) for s in range(4)]
    finals = []
    for run in range(2):
        torch.manual_seed(0)
        m = GraphSAGE(7, 64, 64, 2, 0.3).cuda()
        ts = TrainStep(m, x, ei, seed=99)
        for bt in batches:
            ts.step(*bt)
        assert ts.loss_fn.grad_table.stale
        finals.append(torch.cat([p.detach().flatten() for p in m.parameters()]).clone())
    assert torch.equal(finals[0], finals[1])


def test_full_size_c2_properties():
    """C2 (100k nodes, ~2M edges): unit norms, no NaN/Inf, determinism of the forward, and
    agreement with the oracle run on the same inputs (the oracle finishes in ~1 s here)."""
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 100_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    m, ref = _pair(CFGS[0], seed=0)
    m.eval(), ref.eval()
    with torch.no_grad():
        e1 = m(x, ei)
        e2 = m(x, ei)
        want = ref(x.cpu(), ei.cpu())
    assert torch.equal(e1, e2)
    assert torch.isfinite(e1).all()
    assert torch.allclose(e1.norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)
    assert_close(e1.cpu(), want)


def _torch_sage_gpu(x, ei, conv, n):
    """The oracle's op sequence (index_select -> index_add -> divide -> two linears) as plain torch
    ops ON THE GPU, in float64: an independent implementation that can run at sizes the CPU oracle
    cannot, and accurate enough to judge fp32 sums over 10 M rows."""
    src, dst = ei[0], ei[1]
    agg = torch.zeros(n, x.size(1), device=x.device, dtype=x.dtype).index_add_(0, dst, x[src])
    cnt = torch.bincount(dst, minlength=n).clamp(min=1).to(x.dtype).unsqueeze(1)
    agg = agg / cnt
    return torch.nn.functional.linear(agg, conv.lin_l.weight.double(), conv.lin_l.bias.double()) + \
        torch.nn.functional.linear(x, conv.lin_r.weight.double())


def _hidden_pattern_fp32(m, x, ei, n):
    """h1 > 0 of the fp32 run (the first layer as GraphSAGE.forward runs it in eval mode; the
    kernels are deterministic, so this is the model's own hidden activation).  The float64
    composition takes its ReLU pattern from here: an element whose pre-activation lies within
    rounding of zero may fall on the other side in float64, and each such flip moves a first-layer
    weight gradient by one whole term.  With the pattern fixed the comparison is about arithmetic
    only and every gradient meets the 1e-5 bar."""
    import torch.nn.functional as F
    from graphsage_amd import _lib, ops
    from graphsage_amd.graph import default_cache
    conv = m.convs[0]
    x8 = default_cache().padded(x, 8, keep=True)
    graph = default_cache().get(ei, n)
    agg1 = ops.aggregate_mean(graph, x8)
    h1, _ = ops.project_fwd(agg1, x8, F.pad(conv.lin_l.weight, (0, 1)).detach(), conv.lin_l.bias.detach(),
                            F.pad(conv.lin_r.weight, (0, 1)).detach(), epilogue=_lib.EPI_RELU)
    return h1 > 0


def test_full_size_c4_forward_and_weight_gradients_match_torch_composition():
    """BASELINE C4 (10 M nodes, ~200 M edges, 7->64->64), eval mode: embeddings and the gradients
    of all six parameter tensors for a random linear functional of the embeddings, against the
    same arithmetic composed from torch's own GPU kernels in float64 (index_add, rocBLAS)."""
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 10_000_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    m, _ = _pair(CFGS[0], seed=0)
    m.eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    probe = torch.randn(n, 64, device="cuda", generator=g) / n ** 0.5
    emb = m(x, ei)
    (emb * probe).sum().backward()
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    pattern = _hidden_pattern_fp32(m, x, ei, n)
    pre = _torch_sage_gpu(x.double(), ei, m.convs[0], n)
    flips = (pre > 0) != pattern                              # only where pre is rounding noise
    assert float(pre[flips].abs().max() if bool(flips.any()) else 0.0) <= 1e-5
    h = pre * pattern.double()
    del pre, flips
    want = torch.nn.functional.normalize(_torch_sage_gpu(h, ei, m.convs[1], n), p=2, dim=1)
    del h
    (want * probe.double()).sum().backward()
    err = (emb.detach().double() - want.detach()).abs().max().item()
    assert err <= 2e-6, err                                   # unit-norm rows: absolute = relative
    assert torch.allclose(emb.detach().norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)
    # all six tensors at the oracle bar (the ReLU pattern is the fp32 run's, see above)
    for k, p in m.named_parameters():
        ref = p.grad.double()
        tol = 1e-5 * float(ref.abs().max())
        worst = float((got[k].double() - ref).abs().max())
        assert worst <= tol, (k, worst, tol)


def test_c3_shape_forward_and_weight_gradients_match_float64_composition():
    """C3 (1 M nodes, ~20 M edges, 7->128->128), eval mode: the 128-wide forward and the dX GEMM of
    the backward run on the split-bf16 kernels; embeddings and all parameter gradients against
    the float64 composition, same bars as the C4 test."""
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 1_000_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    m, _ = _pair(CFGS[1], seed=0)
    m.eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    probe = torch.randn(n, 128, device="cuda", generator=g) / n ** 0.5
    emb = m(x, ei)
    (emb * probe).sum().backward()
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    h = _torch_sage_gpu(x.double(), ei, m.convs[0], n) * _hidden_pattern_fp32(m, x, ei, n).double()
    want = torch.nn.functional.normalize(_torch_sage_gpu(h, ei, m.convs[1], n), p=2, dim=1)
    del h
    (want * probe.double()).sum().backward()
    err = (emb.detach().double() - want.detach()).abs().max().item()
    assert err <= 2e-6, err
    for k, p in m.named_parameters():
        ref = p.grad.double()
        tol = 1e-5 * float(ref.abs().max())
        worst = float((got[k].double() - ref).abs().max())
        assert worst <= tol, (k, worst, tol)


def test_c5_shape_forward_matches_float64_composition_at_scale():
    """C5's model (3 layers, 256 wide, eval mode) on 2 M nodes / ~40 M edges: the 256-wide
    projections run on the bf16 matrix pipe with three-way split fp32 operands; embeddings against
    the same arithmetic composed from torch's GPU kernels in float64."""
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 2_000_000
    ei = powerlaw_graph(n, 20, seed=7, device="cuda")
    x = memory_features(n, seed=7, device="cuda")
    m, _ = _pair(CFGS[2], seed=3)
    m.eval()
    with torch.no_grad():
        emb = m(x, ei)
        h = x.double()
        for conv in m.convs[:-1]:
            h = torch.relu(_torch_sage_gpu(h, ei, conv, n))
        want = torch.nn.functional.normalize(_torch_sage_gpu(h, ei, m.convs[-1], n), p=2, dim=1)
    del h
    err = (emb.double() - want).abs().max().item()
    assert err <= 2e-6, err                                   # unit-norm rows: absolute = relative
    assert torch.allclose(emb.norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)


def test_cached_input_aggregation_gives_identical_steps():
    """GraphSAGE(cache_input_aggregation=True): same embeddings and gradients bit for bit, and
    the cache follows in-place edits of the features (version counter)."""
    from graphsage_amd import GraphSAGE
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 3000
    ei = powerlaw_graph(n, 10, seed=2, device="cuda")
    x = memory_features(n, seed=2, device="cuda")
    torch.manual_seed(0)
    a = GraphSAGE(7, 64, 64, 2, 0.0).cuda()
    b = GraphSAGE(7, 64, 64, 2, 0.0, cache_input_aggregation=True).cuda()
    b.load_state_dict(a.state_dict())
    for step in range(3):
        if step == 2:
            x.mul_(0.5)                                    # in place: the cached mean must go
        ea, eb = a(x, ei), b(x, ei)
        assert torch.equal(ea, eb)
        ea.square().sum().backward()
        eb.square().sum().backward()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.equal(pa.grad, pb.grad)
        a.zero_grad(), b.zero_grad()
    xg = x.clone().requires_grad_()
    assert torch.equal(a(xg, ei), b(xg, ei))               # features with grad: never cached


@pytest.mark.parametrize("n,edges", [
    (1, []),                                   # a single node, no edge
    (5, []),                                   # several nodes, no edge at all
    (3, [(0, 0), (1, 1), (2, 2)]),             # self loops only
    (4, [(0, 1), (0, 1), (0, 1), (2, 3)]),     # a triplicated edge
    (130, [(i, 129) for i in range(129)]),     # a star: one row with every other node as neighbour
])
def test_degenerate_graphs_forward_backward_and_loss(n, edges):
    """Tiny and lopsided inputs: eval forward, a full training step's gradients and the loss on a
    one-pair batch, all against the oracle."""
    cfg = dict(in_channels=7, hidden_channels=64, out_channels=64, num_layers=2, dropout=0.0)
    from graphsage_amd import SkipGramLoss
    m, ref = _pair(cfg, seed=n)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 7, generator=g)
    ei = torch.tensor(edges, dtype=torch.int64).reshape(-1, 2).t().contiguous()
    m.train(), ref.train()
    t = torch.randint(0, n, (1,), generator=g)
    c = torch.randint(0, n, (1,), generator=g)
    neg = torch.randint(0, n, (1, 5), generator=g)
    emb = m(x.cuda(), ei.cuda())
    want = ref(x, ei)
    assert_close(emb.detach().cpu(), want.detach())
    loss = SkipGramLoss()(emb, t.cuda(), c.cuda(), neg.cuda())
    ref_loss = O.skipgram_loss(want, t, c, neg)
    assert_close(loss.detach().cpu(), ref_loss.detach())
    loss.backward()
    ref_loss.backward()
    # absolute floor: with one node (or all indices equal) the true gradient is exactly zero --
    # the loss gradient is parallel to the embedding and the L2-norm backward projects it out --
    # so both sides hold rounding noise of ~1e-8 and a floor relative to max|want| means nothing
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        err = (p.grad.cpu().double() - q.grad.double()).abs()
        tol = 1e-4 * q.grad.double().abs() + 1e-6
        assert bool((err <= tol).all()), f"{name}: worst error {float(err.max()):.3e}"
