"""GPU: gsage_segment_gather_hidden1 -- the aggregation of the first hidden layer recomputed from
the input layer's operands -- against what it replaces: the plain segment gather over the stored
activation (PyG: SAGEConv.propagate on h1 = dropout(relu(conv1(x))), graphsage.py:133-135), and
against the same sum in float64 on the CPU."""
import pytest
import torch

from graphsage_amd import _lib

from util import assert_close

pytestmark = pytest.mark.gpu


def _setup(n, fo, deg, seed, feats=7):
    from graphsage_amd.graph import default_cache
    from graphsage_amd.synth import powerlaw_graph
    g = torch.Generator().manual_seed(seed)
    ei = powerlaw_graph(n, deg, seed=seed)
    x8 = torch.zeros(n, 8)
    x8[:, :feats] = torch.rand(n, feats, generator=g) * 2 - 0.5
    wl = torch.zeros(fo, 8)
    wr = torch.zeros(fo, 8)
    wl[:, :feats] = torch.randn(fo, feats, generator=g) / 2
    wr[:, :feats] = torch.randn(fo, feats, generator=g) / 2
    b = torch.randn(fo, generator=g) / 4
    graph = default_cache().get(ei.cuda(), n)
    return ei, x8, wl, b, wr, graph


def _ref64(ei, x8, wl, b, wr, n, act):
    """float64 on the CPU: mean_j act(W_l mean_k x_k + b + W_r x_j)"""
    src, dst = ei[0], ei[1]
    deg = torch.zeros(n, dtype=torch.float64).index_add_(0, dst, torch.ones(ei.size(1), dtype=torch.float64))
    inv = 1.0 / deg.clamp_min(1.0)
    x = x8.double()
    agg1 = torch.zeros(n, 8, dtype=torch.float64).index_add_(0, dst, x[src]) * inv[:, None]
    h1 = act(agg1 @ wl.double().T + b.double() + x @ wr.double().T)
    return torch.zeros(n, wl.size(0), dtype=torch.float64).index_add_(0, dst, h1[src]) * inv[:, None], h1


@pytest.mark.parametrize("n,deg", [(1, 1), (70, 3), (3000, 12), (20000, 40)])
@pytest.mark.parametrize("fo", [64, 128, 256])
def test_eval_matches_the_gather_over_the_stored_activation(n, deg, fo):
    from graphsage_amd import ops
    ei, x8, wl, b, wr, graph = _setup(n, fo, deg, seed=n + fo)
    d = [t.cuda() for t in (x8, wl, b, wr)]
    agg1 = ops.aggregate_mean(graph, d[0])
    h1, _ = ops.project_fwd(agg1, d[0], d[1], d[2], d[3], epilogue=_lib.EPI_RELU)
    want_gpu = ops.aggregate_mean(graph, h1)
    got = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU)
    want, _ = _ref64(ei, x8, wl, b, wr, n, torch.relu)
    assert_close(got, want, what="vs float64")
    assert_close(got, want_gpu, what="vs the plain gather")
    # column-blocked output: the same numbers in the other layout
    gotb = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU, out_blocked=True)
    assert torch.equal(gotb.permute(1, 0, 2).reshape(n, fo), got)


@pytest.mark.parametrize("n,deg", [(1, 1), (333, 5), (5000, 20), (20000, 40)])
@pytest.mark.parametrize("bits", ["sign", "h1"])
@pytest.mark.parametrize("fo", [64, 128, 256])
def test_training_takes_the_zero_pattern_of_the_stored_activation(n, deg, bits, fo):
    """(more than 64 columns: the masks of the further column groups ride behind the 64-byte row)"""
    from graphsage_amd import ops
    p = 0.3
    ei, x8, wl, b, wr, graph = _setup(n, fo, deg, seed=7 * n + fo)
    d = [t.cuda() for t in (x8, wl, b, wr)]
    keep = (torch.rand(n, fo, generator=torch.Generator().manual_seed(n)) >= p).to(torch.uint8)
    agg1 = ops.aggregate_mean(graph, d[0])
    h1, _, sign = ops.project_fwd(agg1, d[0], d[1], d[2], d[3], epilogue=_lib.EPI_RELU_DROPOUT,
                                  keep_mask=keep.cuda(), p=p, want_relu_sign=True)
    want_gpu = ops.aggregate_mean(graph, h1)
    got = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU_DROPOUT, p=p,
                                h1=h1 if bits == "h1" else None, relu_sign=sign if bits == "sign" else None)
    assert_close(got, want_gpu, what="vs the plain gather")
    want, _ = _ref64(ei, x8, wl, b, wr, n, lambda v: torch.relu(v) * keep.double() / (1 - p))
    assert_close(got, want, what="vs float64")


def test_isolated_rows_hub_rows_and_duplicate_edges():
    """a star (one row with 5 000 neighbours: 157 groups), rows without in-edges, a repeated edge"""
    from graphsage_amd import ops
    from graphsage_amd.graph import default_cache
    n = 6000
    hub = torch.stack([torch.arange(1, 5001), torch.zeros(5000, dtype=torch.int64)])
    extra = torch.tensor([[0, 0, 7, 7, 5999], [5500, 5500, 5501, 5501, 5999]])      # duplicates, a self loop
    ei = torch.cat([hub, extra], 1)
    g = torch.Generator().manual_seed(3)
    x8 = torch.zeros(n, 8)
    x8[:, :7] = torch.rand(n, 7, generator=g)
    wl, wr = torch.zeros(64, 8), torch.zeros(64, 8)
    wl[:, :7], wr[:, :7] = torch.randn(64, 7, generator=g), torch.randn(64, 7, generator=g)
    b = torch.randn(64, generator=g)
    graph = default_cache().get(ei.cuda(), n)
    d = [t.cuda() for t in (x8, wl, b, wr)]
    agg1 = ops.aggregate_mean(graph, d[0])
    got = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU)
    want, _ = _ref64(ei, x8, wl, b, wr, n, torch.relu)
    assert_close(got, want)
    assert torch.count_nonzero(got[1:5000]).item() == 0       # leaves of the star have no in-edges


@pytest.mark.parametrize("fo,train", [(64, False), (64, True), (256, False), (128, True)])
def test_rows_around_the_hub_threshold_are_cut_into_pieces(fo, train):
    """rows of 1 023 .. 3 000 neighbours among ordinary ones: above 1 024 a row is summed in pieces
    of 1 024 by other waves and put together afterwards (1 025 = a full piece and one neighbour)"""
    from graphsage_amd import ops
    from graphsage_amd.graph import default_cache
    n, p = 9000, 0.3
    g = torch.Generator().manual_seed(11)
    degs = {5: 1023, 64: 1024, 65: 1025, 700: 2048, 701: 3000, 8999: 1500}
    parts = [torch.stack([torch.randint(0, n, (d,), generator=g), torch.full((d,), r)]) for r, d in degs.items()]
    parts.append(torch.randint(0, n, (2, 40000), generator=g))
    ei = torch.cat(parts, 1)
    x8 = torch.zeros(n, 8)
    x8[:, :7] = torch.rand(n, 7, generator=g) * 2 - 0.5
    wl, wr = torch.zeros(fo, 8), torch.zeros(fo, 8)
    wl[:, :7], wr[:, :7] = torch.randn(fo, 7, generator=g) / 2, torch.randn(fo, 7, generator=g) / 2
    b = torch.randn(fo, generator=g) / 4
    graph = default_cache().get(ei.cuda(), n)
    d = [t.cuda() for t in (x8, wl, b, wr)]
    agg1 = ops.aggregate_mean(graph, d[0])
    if train:
        keep = (torch.rand(n, fo, generator=g) >= p).to(torch.uint8)
        _, _, sign = ops.project_fwd(agg1, d[0], d[1], d[2], d[3], epilogue=_lib.EPI_RELU_DROPOUT,
                                     keep_mask=keep.cuda(), p=p, want_relu_sign=True)
        got = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU_DROPOUT, p=p, relu_sign=sign)
        want, _ = _ref64(ei, x8, wl, b, wr, n, lambda v: torch.relu(v) * keep.double() / (1 - p))
    else:
        got = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU)
        want, _ = _ref64(ei, x8, wl, b, wr, n, torch.relu)
        gotb = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU, out_blocked=True)
        assert torch.equal(gotb.permute(1, 0, 2).reshape(n, fo), got)
    assert_close(got, want, what="vs float64")
    again = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU) if not train else got
    assert torch.equal(again, got)            # piece order is fixed: bitwise reproducible


def test_graph_whose_rows_are_all_hubs():
    """70 nodes, every one with 1 100 .. 1 300 in-edges (duplicates galore): the chunks' own lists
    are empty, every row comes out of the pieces"""
    from graphsage_amd import ops
    from graphsage_amd.graph import default_cache
    n = 70
    g = torch.Generator().manual_seed(5)
    parts = [torch.stack([torch.randint(0, n, (1100 + 3 * r,), generator=g), torch.full((1100 + 3 * r,), r)])
             for r in range(n)]
    ei = torch.cat(parts, 1)
    x8 = torch.zeros(n, 8)
    x8[:, :7] = torch.rand(n, 7, generator=g)
    wl, wr = torch.zeros(128, 8), torch.zeros(128, 8)
    wl[:, :7], wr[:, :7] = torch.randn(128, 7, generator=g) / 2, torch.randn(128, 7, generator=g) / 2
    b = torch.randn(128, generator=g) / 4
    graph = default_cache().get(ei.cuda(), n)
    d = [t.cuda() for t in (x8, wl, b, wr)]
    agg1 = ops.aggregate_mean(graph, d[0])
    got = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU)
    want, _ = _ref64(ei, x8, wl, b, wr, n, torch.relu)
    assert_close(got, want, what="vs float64")


def test_argument_checks():
    from graphsage_amd import ops
    lib = _lib.load()
    assert lib.gsage_segment_gather_hidden1_supported(64, _lib.EPI_RELU_DROPOUT) == 1
    assert lib.gsage_segment_gather_hidden1_supported(128, _lib.EPI_RELU_DROPOUT) == 1
    assert lib.gsage_segment_gather_hidden1_supported(96, _lib.EPI_RELU_DROPOUT) == 0
    assert lib.gsage_segment_gather_hidden1_supported(256, _lib.EPI_RELU) == 1
    assert lib.gsage_segment_gather_hidden1_supported(96, _lib.EPI_RELU) == 0
    assert lib.gsage_segment_gather_hidden1_supported(64, _lib.EPI_L2NORM) == 0
    ei, x8, wl, b, wr, graph = _setup(100, 64, 4, seed=1)
    d = [t.cuda() for t in (x8, wl, b, wr)]
    agg1 = ops.aggregate_mean(graph, d[0])
    with pytest.raises(ValueError):          # training needs the stored zero pattern
        ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU_DROPOUT, p=0.3)
    with pytest.raises(ValueError):          # 7 features padded to 8, nothing else
        ops.aggregate_hidden1(graph, agg1[:, :7].contiguous(), d[0], d[1], d[2], d[3], _lib.EPI_RELU)


@pytest.mark.parametrize("hidden,layers,train", [(64, 2, True), (64, 2, False), (64, 3, True), (256, 3, False),
                                                 (128, 2, True), (256, 3, True)])
def test_model_with_the_recomputed_aggregation_equals_the_plain_path(monkeypatch, hidden, layers, train):
    """GraphSAGE.forward with the second layer's aggregation recomputed (forced on: small graphs
    keep the plain gather by default) against the same model on the plain path: embeddings, and
    in training mode every parameter gradient, with the same injected dropout masks."""
    from graphsage_amd import GraphSAGE, ops
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 4000
    ei = powerlaw_graph(n, 14, seed=5).cuda()
    x = memory_features(n, seed=5).cuda()
    torch.manual_seed(3)
    model = GraphSAGE(7, hidden, 64, layers, 0.3 if train else 0.0).cuda()
    model.train(train)
    g = torch.Generator().manual_seed(9)
    masks = [(torch.rand(n, hidden, generator=g) >= 0.3).to(torch.uint8).cuda() for _ in range(layers - 1)] \
        if train else None
    gout = torch.randn(n, 64, generator=g).cuda()

    def run(min_bytes, recompute):
        monkeypatch.setattr(ops, "HIDDEN1_MIN_BYTES", min_bytes)
        monkeypatch.setattr(ops, "HIDDEN1_RECOMPUTE", recompute)
        model.zero_grad()
        before = ops.HIDDEN1_STATS["used"]
        emb = model(x, ei, keep_masks=masks)
        used = ops.HIDDEN1_STATS["used"] - before
        if train:
            emb.backward(gout)
        return emb.detach().clone(), [p.grad.clone() for p in model.parameters()] if train else [], used

    emb0, grads0, used0 = run(1 << 60, False)
    emb1, grads1, used1 = run(0, True)
    assert used0 == 0 and used1 == 1
    assert_close(emb1, emb0, what="embeddings")
    for a, b, (name, _) in zip(grads1, grads0, model.named_parameters()):
        assert_close(a, b, what=name)


def test_nan_feature_is_zeroed_by_relu_on_both_aggregation_paths(monkeypatch):
    """Pins the library's NaN rule: every forward kernel's ReLU is
    fmaxf(v, 0) = 0 for v = NaN, and the recompute gather (built with -fno-honor-nans) does the
    same, where torch's relu passes the NaN on (graphsage.py:134).  So one NaN input feature
    leaves finite embeddings, identical rows on both paths -- not NaN rows two hops around it."""
    from graphsage_amd import GraphSAGE, ops
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 3000
    ei = powerlaw_graph(n, 10, seed=8).cuda()
    x = memory_features(n, seed=8).cuda()
    x[17, 3] = float("nan")
    torch.manual_seed(1)
    model = GraphSAGE(7, 64, 64, 2, 0.0).cuda().eval()
    monkeypatch.setattr(ops, "HIDDEN1_MIN_BYTES", 0)
    embs = []
    with torch.no_grad():
        for recompute in (True, False):
            monkeypatch.setattr(ops, "HIDDEN1_RECOMPUTE", recompute)
            embs.append(model(x, ei))
    assert torch.isfinite(embs[0]).all() and torch.isfinite(embs[1]).all()
    assert_close(embs[0], embs[1])


@pytest.mark.parametrize("n,deg", [(1, 1), (77, 4), (5003, 20), (20000, 40)])
@pytest.mark.parametrize("fo,train", [(64, False), (64, True), (128, True), (256, True), (256, False)])
def test_rows_written_by_the_input_layer_kernel_equal_the_packed_ones(n, deg, fo, train):
    """gsage_project_fwd_u: the first-layer kernel writes the rows u[j] (operands + its own
    keep-and-positive ballots) while it computes the layer; the gather on that table gives bit for
    bit what the gather on the table packed afterwards from agg / x / the sign bits gives."""
    from graphsage_amd import ops
    p = 0.3
    ei, x8, wl, b, wr, graph = _setup(n, fo, deg, seed=3 * n + fo)
    d = [t.cuda() for t in (x8, wl, b, wr)]
    agg1 = ops.aggregate_mean(graph, d[0])
    epi = _lib.EPI_RELU_DROPOUT if train else _lib.EPI_RELU
    box = {}
    h1, _, sign = ops.project_fwd(agg1, d[0], d[1], d[2], d[3], epilogue=epi, seed=1234 + n, p=p if train else 0.0,
                                  want_relu_sign=True, u_out=box)
    # (one row more than nodes: the first word of the last row is max |u| over the table)
    assert box["u"] is not None and box["u"].shape == (n + 1, 32 if train and fo > 64 else 16)
    want_max = torch.maximum(agg1.abs().max(), d[0].abs().max())
    assert box["u"][n, 0].item() == want_max.item()
    got = ops.aggregate_hidden1_u(graph, box["u"], d[1], d[2], d[3], epi, p=p if train else 0.0)
    want = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], epi, p=p if train else 0.0,
                                 relu_sign=sign if train else None)
    assert torch.equal(got, want)
    assert_close(got, ops.aggregate_mean(graph, h1), what="vs the plain gather over the stored rows")
    # an odd-address keep mask takes another forward kernel: no table, and the box says so
    if train:
        buf = torch.empty(n * fo + 1, dtype=torch.uint8, device="cuda")
        odd = buf[1:].view(n, fo)
        odd.fill_(1)
        box2 = {}
        ops.project_fwd(agg1, d[0], d[1], d[2], d[3], epilogue=epi, keep_mask=odd, p=p, u_out=box2)
        assert box2["u"] is None


@pytest.mark.parametrize("xs,ws,bs", [(1e4, 1.0, 1.0), (1e-5, 1.0, 1.0), (1.0, 1e3, 1e3), (1.0, 1e-4, 10.0),
                                      (300.0, 1e-3, 1e-6), (0.0, 1.0, 1.0)])
@pytest.mark.parametrize("train", [False, True])
def test_fp16_operand_scaling_holds_over_feature_and_weight_scales(xs, ws, bs, train):
    """The recomputed projection runs on fp16 matrix instructions with two-part operands scaled by
    powers of two from max |u| and max |W'|: features of size 1e4 or 1e-5, weights of size 1e3 or
    1e-4, a bias far above or below the weights, all-zero features -- the result stays within the
    oracle bar of the float64 value (which an unscaled fp16 operand would leave at once: 1e4 * 8
    terms overflows, 1e-5 is subnormal)."""
    from graphsage_amd import ops
    n, fo, p = 6000, 64, 0.3
    ei, x8, wl, b, wr, graph = _setup(n, fo, 12, seed=21)
    x8, wl, wr, b = x8 * xs, wl * ws, wr * ws, b * bs
    d = [t.cuda() for t in (x8, wl, b, wr)]
    agg1 = ops.aggregate_mean(graph, d[0])
    if train:
        keep = (torch.rand(n, fo, generator=torch.Generator().manual_seed(n)) >= p).to(torch.uint8)
        box = {}
        _, _, sign = ops.project_fwd(agg1, d[0], d[1], d[2], d[3], epilogue=_lib.EPI_RELU_DROPOUT,
                                     keep_mask=keep.cuda(), p=p, want_relu_sign=True, u_out=box)
        got = ops.aggregate_hidden1_u(graph, box["u"], d[1], d[2], d[3], _lib.EPI_RELU_DROPOUT, p=p)
        want, _ = _ref64(ei, x8, wl, b, wr, n, lambda v: torch.relu(v) * keep.double() / (1 - p))
    else:
        got = ops.aggregate_hidden1(graph, agg1, d[0], d[1], d[2], d[3], _lib.EPI_RELU)
        want, _ = _ref64(ei, x8, wl, b, wr, n, torch.relu)
    assert torch.isfinite(got).all()
    assert_close(got, want, what="vs float64")
