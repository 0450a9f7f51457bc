"""GPU: gsage_skipgram_fwd_noalloc (the one entry point that allocates nothing: its per-workgroup
partials live in a file-scope device array of the library) against gsage_skipgram_fwd with a
caller workspace, through the C ABI.  Same kernels => bit-identical coef and stats.  Two calls in
a row, the second on a smaller input (the array keeps the first call's contents)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(n, d, b, k, seed):
    g = torch.Generator().manual_seed(seed)
    emb = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).cuda()
    t = torch.randint(0, n, (b,), generator=g).cuda()
    c = torch.randint(0, n, (b,), generator=g).cuda()
    neg = torch.randint(0, n, (b, k), generator=g).cuda()
    return emb, t, c, neg


def _with_workspace(lib, _lib, emb, t, c, neg, scale=5.0):
    n, d = emb.shape
    b, k = neg.shape
    coef = torch.empty(b, 1 + k, device="cuda")
    stats = torch.empty(8, device="cuda")
    ws_bytes = lib.gsage_skipgram_workspace_bytes(b)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    rc = lib.gsage_skipgram_fwd(emb.data_ptr(), n, d, t.data_ptr(), c.data_ptr(), neg.data_ptr(), b, k,
                                scale, float(b), float(b * k), coef.data_ptr(), stats.data_ptr(),
                                ws.data_ptr(), ws_bytes, _lib.stream_handle())
    assert rc == 0
    return coef, stats


def _noalloc(lib, _lib, emb, t, c, neg, coef, stats, scale=5.0):
    n, d = emb.shape
    b, k = neg.shape
    return lib.gsage_skipgram_fwd_noalloc(emb.data_ptr(), n, d, t.data_ptr(), c.data_ptr(),
                                          neg.data_ptr(), b, k, scale, float(b), float(b * k),
                                          coef.data_ptr(), stats.data_ptr(), _lib.stream_handle())


def test_noalloc_forward_equals_workspace_forward_twice_in_a_row():
    from graphsage_amd import _lib
    lib = _lib.load()
    big = _inputs(50_000, 64, 70_001, 5, seed=1)      # > 2048 workgroups' worth: the grid cap is hit
    small = _inputs(3_000, 64, 37, 5, seed=2)         # 3 workgroups: most of the array keeps call 1's sums
    want_big = _with_workspace(lib, _lib, *big)
    want_small = _with_workspace(lib, _lib, *small)
    # outputs allocated BEFORE the calls under test; the calls themselves get pointers only
    outs = [(torch.full((x[1].numel(), 6), float("nan"), device="cuda"),
             torch.full((8,), float("nan"), device="cuda")) for x in (big, small)]
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated()
    assert _noalloc(lib, _lib, *big, *outs[0]) == 0
    assert _noalloc(lib, _lib, *small, *outs[1]) == 0     # second call, smaller input
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == mem0
    for (coef, stats), (wcoef, wstats) in zip(outs, (want_big, want_small)):
        assert torch.equal(coef, wcoef)                   # bit-exact: same kernels, same order
        assert torch.equal(stats, wstats)
        assert stats[7].item() == 0 and torch.isfinite(stats[:7]).all()
    # and against the CPU oracle (the bar of the other loss tests: 1e-5 relative)
    from oracle import sage_oracle as O
    emb, t, c, neg = (x.cpu() for x in small)
    ref, _ = O.skipgram_loss_with_details(emb, t, c, neg, 5.0)
    assert outs[1][1][0].item() == pytest.approx(ref.item(), rel=1e-5)


def test_noalloc_other_widths_and_bad_index_flag():
    from graphsage_amd import _lib
    lib = _lib.load()
    for n, d, b, k, seed in [(2000, 128, 1000, 3, 3), (500, 30, 77, 7, 4)]:   # vec<2> and generic kernel
        x = _inputs(n, d, b, k, seed)
        want = _with_workspace(lib, _lib, *x)
        coef, stats = torch.empty(b, 1 + k, device="cuda"), torch.empty(8, device="cuda")
        assert _noalloc(lib, _lib, *x, coef, stats) == 0
        assert torch.equal(coef, want[0]) and torch.equal(stats, want[1])
    emb, t, c, neg = _inputs(100, 64, 9, 2, 5)
    c[4] = 100                                            # out of range: flagged, not fetched
    coef, stats = torch.empty(9, 3, device="cuda"), torch.empty(8, device="cuda")
    assert _noalloc(lib, _lib, emb, t, c, neg, coef, stats) == 0
    assert int(stats[7].item()) & _lib.FLAG_INDEX_OUT_OF_RANGE
    # the next call starts clean although the array still holds the flagged partials
    c[4] = 5
    assert _noalloc(lib, _lib, emb, t, c, neg, coef, stats) == 0
    assert stats[7].item() == 0


def test_noalloc_rejects_sizes_beyond_its_maximum():
    from graphsage_amd import _lib
    lib = _lib.load()
    cap = lib.gsage_skipgram_noalloc_max_pairs()
    assert cap == 1 << 24
    emb, t, c, neg = _inputs(100, 64, 8, 5, 6)
    coef, stats = torch.zeros(8, 6, device="cuda"), torch.full((8,), -1.0, device="cuda")
    rc = lib.gsage_skipgram_fwd_noalloc(emb.data_ptr(), 100, 64, t.data_ptr(), c.data_ptr(),
                                        neg.data_ptr(), cap + 1, 5, 5.0, 8.0, 40.0, coef.data_ptr(),
                                        stats.data_ptr(), _lib.stream_handle())
    assert rc == -5 and b"no-alloc" in lib.gsage_error_string(rc)
    torch.cuda.synchronize()
    assert (stats == -1.0).all()                          # nothing was launched
    with pytest.raises(ValueError):
        _lib.check(rc, "gsage_skipgram_fwd_noalloc")
