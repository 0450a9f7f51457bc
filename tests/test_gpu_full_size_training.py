"""GPU: the TRAINING path at the sizes bench.py times it (BASELINE configs 4 and 3).

The reference's hidden layers are relu -> dropout(0.3) (src/model/graphsage.py:132-135), so the
benchmark runs the training variants of the kernels: the in-kernel dropout hash and the sign
ballots of the first-layer projection, the recompute gather with the keep-and-positive bits in
its rows (hidden1_gather_k<2,true> at 64 columns, <4,true> with 128-byte rows at 128), and the
fused layer-2 backward.  The smaller parity tests stop at 20 000 rows; these run the same checks
at 10 M rows (C4) and 1 M rows (C3), where 32-bit offsets and word counts are at their largest.

The comparison is the float64 composition of torch's own GPU kernels (index_add, rocBLAS) with the
SAME dropout mask, and with the ReLU pattern taken from the fp32 run: an element whose
pre-activation lies within rounding of zero may fall on the other side in float64, and each such
flip moves a first-layer weight gradient by one whole term; with the pattern fixed the comparison
is about arithmetic only and the first layer's gradients meet the same 1e-5 bar as the rest."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from graphsage_amd import _lib

pytestmark = pytest.mark.gpu

P = 0.3


def _hash_keep_mask(seed, n_rows, n_cols, p, device):
    """graphsage_amd/rng.py::dropout_keep_mask (the host mirror of the kernel's hash) in torch on
    the GPU: 32-bit arithmetic carried in int64 and masked, chunked over rows."""
    seed &= 0xFFFFFFFFFFFFFFFF
    m32 = 0xFFFFFFFF
    s_lo, s_hi = seed & m32, seed >> 32
    out = torch.empty(n_rows, n_cols, dtype=torch.uint8, device=device)
    step = 1 << 20
    for r0 in range(0, n_rows, step):
        r1 = min(n_rows, r0 + step)
        idx = torch.arange(r0 * n_cols, r1 * n_cols, dtype=torch.int64, device=device)
        lo, hi = idx & m32, idx >> 32
        h = s_lo ^ ((lo * 0x9E3779B1) & m32)
        h = h ^ ((s_hi + hi * 0x85EBCA77) & m32)
        h = h ^ (h >> 16)
        h = (h * 0x85EBCA6B) & m32
        h = h ^ (h >> 13)
        h = (h * 0xC2B2AE35) & m32
        h = h ^ (h >> 16)
        u = (h >> 8).to(torch.float32) * (1.0 / 16777216.0)
        out[r0:r1] = (u >= p).view(r1 - r0, n_cols)
    return out


def _sign_bits_to_rows(sign, n, fo, r0, r1):
    """Decode rows r0..r1 (multiples of 4) of the buffer gsage_project_fwd_ex leaves
    (gsage_relu_sign_bytes): per 4 rows and 64-column group four 64-bit words, word cc bit
    16 (row & 3) + q = column 64 g + 4 q + cc.  -> bool [r1 - r0, fo]"""
    ngrp4 = (n + 3) // 4
    groups = fo // 64
    cols = []
    sh = (16 * torch.arange(4, device=sign.device).view(1, 1, 4, 1)
          + torch.arange(16, device=sign.device).view(1, 1, 1, 16))            # [1, 1, jj, q]
    for g in range(groups):
        w = sign[g * ngrp4 * 4:(g + 1) * ngrp4 * 4].view(ngrp4, 4)[r0 // 4:r1 // 4]   # [R/4, cc]
        bits = (w.view(-1, 4, 1, 1) >> sh) & 1                                  # [R/4, cc, jj, q]
        cols.append(bits.permute(0, 2, 3, 1).reshape(r1 - r0, 64).bool())       # column = 4 q + cc
    return torch.cat(cols, 1)


def _conv64(x, ei, conv, n):
    """SAGEConv in float64 out of torch's GPU ops: index_select -> index_add -> / count -> linears."""
    src, dst = ei[0], ei[1]
    agg = torch.zeros(n, x.size(1), device=x.device, dtype=x.dtype).index_add_(0, dst, x[src])
    cnt = torch.bincount(dst, minlength=n).clamp(min=1).to(x.dtype).unsqueeze(1)
    agg = agg / cnt
    return F.linear(agg, conv.lin_l.weight.double(), conv.lin_l.bias.double()) + \
        F.linear(x, conv.lin_r.weight.double())


def _first_layer_fp32(model, x, graph, keep=None, seed=0):
    """The first layer exactly as GraphSAGE.forward runs it (deterministic kernels: bit-identical
    to the model's own hidden activation) -> (h1, sign buffer)."""
    from graphsage_amd import ops
    from graphsage_amd.graph import default_cache
    conv = model.convs[0]
    x8 = default_cache().padded(x, 8, keep=True)
    w_l, w_r = F.pad(conv.lin_l.weight, (0, 1)).detach(), F.pad(conv.lin_r.weight, (0, 1)).detach()
    agg1 = ops.aggregate_mean(graph, x8)
    h1, _, sign = ops.project_fwd(agg1, x8, w_l, conv.lin_l.bias.detach(), w_r,
                                  epilogue=_lib.EPI_RELU_DROPOUT, keep_mask=keep, seed=seed, p=P,
                                  want_relu_sign=True)
    return h1, sign


def _run_model(model, x, ei, probe, keep_masks, recompute, monkeypatch):
    from graphsage_amd import ops
    monkeypatch.setattr(ops, "HIDDEN1_RECOMPUTE", recompute)
    model.zero_grad()
    before = ops.HIDDEN1_STATS["used"]
    emb = model(x, ei, keep_masks=keep_masks)
    used = ops.HIDDEN1_STATS["used"] - before
    (emb * probe).sum().backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    return emb.detach(), grads, used


def _float64_composition(model, x, ei, n, pattern, probe):
    """embeddings and parameter gradients in float64 with the first layer's zero pattern given
    (kept AND positive in the fp32 run): h = pre * pattern / (1 - p)."""
    pre = _conv64(x.double(), ei, model.convs[0], n)
    h = pre * pattern.to(torch.float64) / (1.0 - P)
    del pre
    want = F.normalize(_conv64(h, ei, model.convs[1], n), p=2, dim=1)
    del h
    (want * probe.double()).sum().backward()
    grads = {k: p.grad.double().clone() for k, p in model.named_parameters()}
    model.zero_grad()
    return want.detach(), grads


def _check(emb, grads, want, want_grads, what, emb_tol=2e-6, grad_tol=1e-5):
    err = (emb.double() - want).abs().max().item()
    assert err <= emb_tol, (what, "embeddings", err)              # unit rows: absolute = relative
    for k, ref in want_grads.items():
        tol = grad_tol * float(ref.abs().max())
        worst = float((grads[k].double() - ref).abs().max())
        assert worst <= tol, (what, k, worst, tol)


@pytest.fixture(scope="module")
def c4():
    from graphsage_amd.graph import default_cache
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 10_000_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    graph = default_cache().get(ei, n)
    g = torch.Generator(device="cuda").manual_seed(1)
    probe = torch.randn(n, 64, device="cuda", generator=g) / n ** 0.5
    yield n, ei, x, graph, probe
    del ei, x, graph, probe
    torch.cuda.empty_cache()


def _model(hidden, seed=0):
    from graphsage_amd import GraphSAGE
    torch.manual_seed(seed)
    return GraphSAGE(7, hidden, hidden, 2, P).cuda().train()


def test_c4_training_step_with_injected_masks_recompute_on_off_and_float64(c4, monkeypatch):
    """(a) GraphSAGE(7,64,64,2,0.3).train() on the 10 M-node graph with an injected keep mask:
    embeddings and all six parameter gradients with the layer-2 aggregation recomputed
    (hidden1_gather_k<2,true>) and gathered (GSAGE_HIDDEN1_RECOMPUTE=0), against each other and
    both against the float64 composition with the same mask."""
    n, ei, x, graph, probe = c4
    model = _model(64)
    g = torch.Generator(device="cuda").manual_seed(5)
    keep = (torch.rand(n, 64, device="cuda", generator=g) >= P).to(torch.uint8)
    emb1, grads1, used1 = _run_model(model, x, ei, probe, [keep], True, monkeypatch)
    emb0, grads0, used0 = _run_model(model, x, ei, probe, [keep], False, monkeypatch)
    assert used1 == 1 and used0 == 0
    h1, sign = _first_layer_fp32(model, x, graph, keep=keep)
    pattern = h1 > 0
    # the sign bits the backward and the recompute gather read are exactly "kept and positive"
    assert torch.equal(_sign_bits_to_rows(sign, n, 64, 0, 1 << 20), pattern[:1 << 20])
    assert torch.equal(_sign_bits_to_rows(sign, n, 64, 1 << 23, (1 << 23) + (1 << 20)),
                       pattern[1 << 23:(1 << 23) + (1 << 20)])
    assert torch.equal(_sign_bits_to_rows(sign, n, 64, n - (1 << 16), n), pattern[n - (1 << 16):])
    del h1, sign
    want, want_grads = _float64_composition(model, x, ei, n, pattern, probe)
    _check(emb1, grads1, want, want_grads, "recomputed aggregation")
    _check(emb0, grads0, want, want_grads, "plain gather")
    # recompute on vs off: same zero pattern, values equal to fp32 rounding
    assert (emb1 - emb0).abs().max().item() <= 3e-6
    for k in grads0:
        assert float((grads1[k] - grads0[k]).abs().max()) <= 1e-5 * float(grads0[k].abs().max()), k


def test_c4_training_step_with_the_in_kernel_dropout_hash(c4, monkeypatch):
    """(b) the form the benchmark runs: no injected mask, the projection kernel draws it from its
    counter hash.  The mask the kernel used is rebuilt from the host mirror of the hash
    (graphsage_amd/rng.py, here in torch on the GPU and checked against the numpy mirror on a
    slice of rows beyond 2^23), the sign-bit buffer is checked against h1 > 0 there, and the whole
    forward + backward against the float64 composition with that mask."""
    from graphsage_amd import rng
    n, ei, x, graph, probe = c4
    model = _model(64, seed=1)
    seed = 0x1234_5678_9ABC_DEF1
    monkeypatch.setattr(model, "_dropout_seed", lambda layer=0: seed)
    emb1, grads1, used1 = _run_model(model, x, ei, probe, None, True, monkeypatch)
    emb0, grads0, used0 = _run_model(model, x, ei, probe, None, False, monkeypatch)
    assert used1 == 1 and used0 == 0
    keep = _hash_keep_mask(seed, n, 64, P, "cuda")
    r0, rows = 1 << 23, 1 << 16
    # the host mirror, row slice [2^23, 2^23 + 2^16): element indices start at 2^29
    part = rng.dropout_keep_mask_range(seed, r0 * 64, rows * 64, P).reshape(rows, 64)
    assert np.array_equal(part, keep[r0:r0 + rows].cpu().numpy())
    h1, sign = _first_layer_fp32(model, x, graph, keep=None, seed=seed)
    # what the kernel dropped is what the mirror says: no kept-out element is non-zero, and the
    # kept elements are exactly the positive pre-activations (checked through the float64 run below)
    assert not bool((h1[~keep.bool()] != 0).any())
    pattern = h1 > 0
    hi = slice(r0, r0 + (1 << 20))
    assert torch.equal(_sign_bits_to_rows(sign, n, 64, r0, r0 + (1 << 20)), pattern[hi])
    assert 0.69 < keep[hi].float().mean().item() < 0.71
    del h1, sign
    want, want_grads = _float64_composition(model, x, ei, n, pattern, probe)
    # pattern must be (pre64 > 0) & keep except where pre is within rounding of zero
    pre = _conv64(x.double(), ei, model.convs[0], n)
    disagree = ((pre > 0) & keep.bool()) != pattern
    assert float(pre[disagree].abs().max() if bool(disagree.any()) else 0.0) <= 1e-5
    assert int(disagree.sum()) <= n * 64 // 100_000
    del pre, disagree
    _check(emb1, grads1, want, want_grads, "recomputed aggregation, hash dropout")
    _check(emb0, grads0, want, want_grads, "plain gather, hash dropout")


def test_c3_training_step_128_wide_with_128_byte_rows(monkeypatch):
    """(c) C3: 1 M nodes, 7 -> 128 -> 128, dropout 0.3: hidden1_gather_k<4,true> (the masks of
    columns 64..127 ride behind the 64-byte row) against the plain gather and float64."""
    from graphsage_amd.graph import default_cache
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 1_000_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    graph = default_cache().get(ei, n)
    g = torch.Generator(device="cuda").manual_seed(2)
    probe = torch.randn(n, 128, device="cuda", generator=g) / n ** 0.5
    model = _model(128, seed=2)
    seed = 0xC3C3_0000_1111_2222
    monkeypatch.setattr(model, "_dropout_seed", lambda layer=0: seed)
    for keep_in in (None, "injected"):
        if keep_in is None:
            keep = _hash_keep_mask(seed, n, 128, P, "cuda")
            masks = None
        else:
            keep = (torch.rand(n, 128, device="cuda", generator=g) >= P).to(torch.uint8)
            masks = [keep]
        emb1, grads1, used1 = _run_model(model, x, ei, probe, masks, True, monkeypatch)
        emb0, grads0, used0 = _run_model(model, x, ei, probe, masks, False, monkeypatch)
        assert used1 == 1 and used0 == 0
        h1, sign = _first_layer_fp32(model, x, graph, keep=masks[0] if masks else None, seed=seed)
        assert not bool((h1[~keep.bool()] != 0).any())
        pattern = h1 > 0
        assert torch.equal(_sign_bits_to_rows(sign, n, 128, 0, n), pattern)
        del h1, sign
        want, want_grads = _float64_composition(model, x, ei, n, pattern, probe)
        _check(emb1, grads1, want, want_grads, f"recomputed, mask {keep_in}")
        _check(emb0, grads0, want, want_grads, f"plain gather, mask {keep_in}")


def test_training_recompute_when_the_forward_kernel_left_no_sign_bits(monkeypatch):
    """A keep mask at an odd address sends the first-layer forward to a kernel that leaves no sign
    bits (the flag word behind them says so): the recompute gather must then take the zero pattern
    from the stored activation, which the layer hands it, instead of reading through a null
    pointer."""
    from graphsage_amd import GraphSAGE, ops
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 5000
    ei = powerlaw_graph(n, 14, seed=5).cuda()
    x = memory_features(n, seed=5).cuda()
    torch.manual_seed(3)
    model = GraphSAGE(7, 64, 64, 2, P).cuda().train()
    g = torch.Generator().manual_seed(9)
    keep = (torch.rand(n, 64, generator=g) >= P).to(torch.uint8)
    buf = torch.empty(n * 64 + 1, dtype=torch.uint8, device="cuda")
    odd = buf[1:].view(n, 64)
    odd.copy_(keep)
    assert odd.data_ptr() % 2 == 1 and odd.is_contiguous()
    gout = torch.randn(n, 64, generator=g).cuda()
    monkeypatch.setattr(ops, "HIDDEN1_MIN_BYTES", 0)
    res = []
    for mask, recompute in ((odd, True), (keep.cuda(), True), (keep.cuda(), False)):
        monkeypatch.setattr(ops, "HIDDEN1_RECOMPUTE", recompute)
        model.zero_grad()
        before = ops.HIDDEN1_STATS["used"]
        emb = model(x, ei, keep_masks=[mask])
        assert (ops.HIDDEN1_STATS["used"] - before) == (1 if recompute else 0)
        emb.backward(gout)
        res.append((emb.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    from util import assert_close
    for other in res[1:]:
        assert_close(res[0][0], other[0], what="embeddings")
        for a, b in zip(res[0][1], other[1]):
            assert_close(a, b, what="gradient")


@pytest.mark.parametrize("hidden,n", [(64, (1 << 26) - 8), (128, (1 << 25) - 8), (64, (1 << 26) + 8)])
def test_row_table_at_the_top_of_its_32_bit_offsets(monkeypatch, hidden, n):
    """The recompute gather and gsage_hidden1_wgrad address the 64-byte rows u[j] by 32-bit byte offsets, which
    is why they take graphs of fewer than 2^26 nodes.  A graph of 2^26 - 8 nodes whose few edges and pairs sit
    at both ends of the id range: embeddings of the named rows and all six parameter gradients with the
    recompute gather and the hand-over of the input layer's weight gradients on, against the plain gather and
    the transposed-gather backward (64-bit addressing throughout) on the same dropout seeds.  128 columns wide
    the rows take 128 bytes in training (the further column groups' masks ride behind them) and the limit is
    2^25 nodes; there is no hand-over at that width.  Third case: 8 nodes PAST the limit, where both must step
    aside by themselves (plain gather, row-list backward with the masked transposed gather) and the result is
    compared with the dense backward.  The reference semantics of all of it are graphsage.py:132-138 and
    autograd of it."""
    from graphsage_amd import GraphSAGE, SkipGramLoss, ops
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(11)
    top = lambda k: n - 1 - torch.randint(0, 6000, (k,), device=dev, generator=g)        # noqa: E731
    low = lambda k: torch.randint(0, 6000, (k,), device=dev, generator=g)                # noqa: E731
    mid = lambda k: (n // 2) + torch.randint(0, 6000, (k,), device=dev, generator=g)     # noqa: E731
    dst = torch.cat([top(3000), low(3000), mid(3000)]).repeat_interleave(12)
    src = torch.cat([top(36000), low(36000), top(18000), mid(18000)])[torch.randperm(108000, device=dev, generator=g)]
    ei = torch.stack([src, dst])
    x = torch.rand(n, 7, device=dev, generator=g)
    b = 2000
    t = torch.cat([top(b // 2), low(b // 2)])
    c = torch.cat([low(b // 2), top(b // 2)])
    neg = torch.cat([top(b * 2), mid(b), low(b * 2)]).view(b, 5)
    named = torch.unique(torch.cat([t, c, neg.flatten()]))
    results = []
    for on in (True, False):
        monkeypatch.setattr(ops, "HIDDEN1_MIN_BYTES", 0 if on else (1 << 62))
        monkeypatch.setattr(ops, "HIDDEN1_WGRAD", on)
        monkeypatch.setattr(ops, "SKIP_ZERO_ROWS", on)
        torch.manual_seed(0)
        m = GraphSAGE(7, hidden, hidden, 2, P).to(dev).train()
        m.set_dropout_stream(1234567, 0)
        used = (ops.HIDDEN1_STATS["used"], ops.WGRAD_STATS["used"], ops.LIVE_STATS["used"])
        emb = m(x, ei)
        loss = SkipGramLoss()(emb, t, c, neg)
        loss.backward()
        inside = n < (1 << 26) // (hidden // 64)
        if on:
            assert ops.HIDDEN1_STATS["used"] == used[0] + (1 if inside else 0)
            assert ops.WGRAD_STATS["used"] == used[1] + (1 if hidden == 64 and inside else 0)
            if hidden == 64:
                assert ops.LIVE_STATS["used"] == used[2] + 1
        else:
            assert ops.HIDDEN1_STATS["used"] == used[0] and ops.WGRAD_STATS["used"] == used[1]
        results.append((emb.detach()[named].cpu(), float(loss.detach()), [p.grad.detach().cpu().clone() for p in m.parameters()]))
        del emb, loss, m
        torch.cuda.empty_cache()
    (e1, l1, g1), (e0, l0, g0) = results
    scale = float(e0.abs().max())
    assert float((e1 - e0).abs().max()) <= 1e-5 * scale
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    for a, w in zip(g1, g0):
        assert float((a - w).abs().max()) <= 2e-5 * float(w.abs().max()) + 1e-12
