"""GPU: the 256-column forward projection on the fp16 matrix pipe (two-way split of every fp32
operand, gemm_half_k) and the range hand-over between layers (gsage_project_fwd_range).  The
reference computes these layers in fp32 (SAGEConv's two F.linear, graphsage.py:133-142 with
hidden_dim 256, config 5 of BASELINE.json); the bar is the same 1e-5 as for every other kernel.
What is specific to this kernel is its RANGE handling, so every case here is built to leave the
comfortable range: rows far below the largest one, a bound that is too low, non-finite operands,
magnitudes fp16 cannot be scaled to, and more doubtful tiles than the redo list holds."""
import os

import pytest
import torch
import torch.nn.functional as F

from graphsage_amd import _lib

from util import assert_close

pytestmark = pytest.mark.gpu

forced = os.environ.get("GSAGE_WIDE_GEMM", "")[:1] in ("f", "s")   # a forced other family: no fp16 kernel


def _bits(v: float) -> int:
    return int(torch.tensor([v], dtype=torch.float32).view(torch.int32).item())


def _case(n, f=256, seed=0, scale_rows=None):
    g = torch.Generator().manual_seed(seed)
    agg = torch.randn(n, f, generator=g)
    x = torch.randn(n, f, generator=g)
    if scale_rows is not None:
        agg *= scale_rows[:, None]
        x *= scale_rows[:, None]
    wl = torch.randn(f, f, generator=g) / f ** 0.5
    wr = torch.randn(f, f, generator=g) / f ** 0.5
    b = torch.randn(f, generator=g)
    return agg, x, wl, b, wr


def _ref64(agg, x, wl, b, wr, epi):
    pre = agg.double() @ wl.double().T + b.double() + x.double() @ wr.double().T
    if epi == _lib.EPI_RELU:
        return torch.relu(pre)
    if epi == _lib.EPI_L2NORM:
        return F.normalize(pre, dim=1)
    return pre


def _run(agg, x, wl, b, wr, epi, amax_in=None, want_amax=False):
    from graphsage_amd.ops import project_fwd_blocked
    d = [t.cuda() for t in (agg, x, wl, b, wr)]
    ain = None if amax_in is None else torch.tensor([_bits(amax_in)], dtype=torch.int32, device="cuda")
    aout = torch.full((1,), 12345, dtype=torch.int32, device="cuda") if want_amax else None
    out, _ = project_fwd_blocked(*d, epi, False, False, amax_in=ain, amax_out=aout)
    return (out, aout) if want_amax else out


def _rowwise_close(got, want, rtol=1e-5):
    """every row to 1e-5 of ITS OWN largest element (a tiny row must not be judged against the
    largest row of the matrix)"""
    got, want = got.detach().cpu().double(), want.double()
    scale = want.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    err = ((got - want).abs() / scale).amax(dim=1)
    worst = int(err.argmax())
    assert float(err[worst]) <= rtol, f"row {worst}: relative error {float(err[worst]):.3e}"


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RELU, _lib.EPI_L2NORM])
@pytest.mark.parametrize("n", [1, 255, 256, 257, 5000])
def test_forward_with_and_without_a_bound(n, epi):
    agg, x, wl, b, wr = _case(n, seed=n)
    want = _ref64(agg, x, wl, b, wr, epi)
    out0 = _run(agg, x, wl, b, wr, epi)                              # the library finds max |A|
    m = float(torch.maximum(agg.abs().max(), x.abs().max()))
    out1, amax = _run(agg, x, wl, b, wr, epi, amax_in=m, want_amax=True)
    assert_close(out0, want, what="own maximum")
    assert_close(out1, want, what="bound handed in")
    assert torch.equal(out0, out1)           # the same power of two either way: the same bits
    # the published maximum is the maximum of what was stored, bit for bit
    assert int(amax.item()) == int(out1.abs().max().view(torch.int32).item())
    # a generous bound (x 1.9: the same binade or the next) costs at most one bit of headroom
    out2 = _run(agg, x, wl, b, wr, epi, amax_in=1.9 * m)
    assert_close(out2, want, what="loose bound")


def test_first_layer_publishes_its_maximum_and_the_chain_uses_it():
    """8 -> 256 (first-layer MFMA kernel, blocked output) -> 256 -> 256 with the bound it left."""
    from graphsage_amd.ops import project_fwd_blocked
    n = 3000
    g = torch.Generator().manual_seed(3)
    agg8, x8 = torch.randn(n, 8, generator=g), torch.randn(n, 8, generator=g)
    w1l, w1r, b1 = torch.randn(256, 8, generator=g), torch.randn(256, 8, generator=g), torch.randn(256, generator=g)
    amax = torch.zeros(2, dtype=torch.int32, device="cuda")
    h, _ = project_fwd_blocked(agg8.cuda(), x8.cuda(), w1l.cuda(), b1.cuda(), w1r.cuda(), _lib.EPI_RELU,
                               False, True, amax_out=amax[0:1])
    h_rows = h.permute(1, 0, 2).reshape(n, 256)
    assert int(amax[0].item()) == int(h_rows.max().view(torch.int32).item())
    want1 = torch.relu(agg8.double() @ w1l.double().T + b1.double() + x8.double() @ w1r.double().T)
    assert_close(h_rows, want1)
    _, _, wl, b, wr = _case(4, seed=9)
    agg2 = (h * 0.5).contiguous()             # any operand below the bound, blocked like h
    out, _ = project_fwd_blocked(agg2, h, wl.cuda(), b.cuda(), wr.cuda(), _lib.EPI_RELU, True, True,
                                 amax_in=amax[0:1], amax_out=amax[1:2])
    out_rows = out.permute(1, 0, 2).reshape(n, 256)
    want2 = _ref64(0.5 * h_rows.cpu(), h_rows.cpu(), wl, b, wr, _lib.EPI_RELU)
    assert_close(out_rows, want2)
    assert int(amax[1].item()) == int(out_rows.max().view(torch.int32).item())


@pytest.mark.skipif(forced, reason="another kernel family is forced")
def test_rows_far_below_the_largest_keep_their_accuracy():
    """Rows 2^-30 .. 2^-60 below the largest row cannot be held in scaled fp16: their tiles are
    recomputed on the bf16 kernel.  No bias, so a tiny row's output is tiny too and is judged by
    its own size."""
    n = 2000
    g = torch.Generator().manual_seed(5)
    s = torch.ones(n)
    s[100:140] = 2.0 ** -30
    s[777] = 2.0 ** -60
    s[1999] = 2.0 ** -45
    s[300:310] = 2.0 ** -17          # still inside what two fp16 parts carry
    agg, x, wl, b, wr = _case(n, seed=5, scale_rows=s)
    b.zero_()
    want = _ref64(agg, x, wl, b, wr, _lib.EPI_NONE)
    out = _run(agg, x, wl, b, wr, _lib.EPI_NONE)
    _rowwise_close(out, want)
    out = _run(agg, x, wl, b, wr, _lib.EPI_NONE, amax_in=float(x.abs().max()))
    _rowwise_close(out, want)


@pytest.mark.skipif(forced, reason="another kernel family is forced")
def test_a_bound_that_is_too_low_is_noticed():
    """amax_in 2^12 below the true maximum: scaled rows overflow fp16; the kernel notices (row
    maximum beyond 65504 after scaling) and the tiles are recomputed."""
    n = 1500
    agg, x, wl, b, wr = _case(n, seed=6)
    want = _ref64(agg, x, wl, b, wr, _lib.EPI_RELU)
    out = _run(agg, x, wl, b, wr, _lib.EPI_RELU, amax_in=float(x.abs().max()) / 4096)
    assert_close(out, want)
    # ... and one outlier row above an otherwise honest bound
    x[1234] *= 300.0
    want = _ref64(agg, x, wl, b, wr, _lib.EPI_RELU)
    out = _run(agg, x, wl, b, wr, _lib.EPI_RELU, amax_in=float(x[:1000].abs().max()))
    _rowwise_close(out, want)


@pytest.mark.parametrize("big", [1e30, 1e-30])
def test_magnitudes_outside_the_scalable_range(big):
    agg, x, wl, b, wr = _case(700, seed=7)
    agg, x = agg * big, x * big
    b = b * big
    want = _ref64(agg, x, wl, b, wr, _lib.EPI_NONE)
    out = _run(agg, x, wl, b, wr, _lib.EPI_NONE)
    assert_close(out, want)


def test_non_finite_operands_behave_as_in_fp32():
    """An Inf or NaN in a row poisons that row of the output (as it does in fp32) and nothing else."""
    n = 900
    agg, x, wl, b, wr = _case(n, seed=8)
    x[17, 3] = float("inf")
    agg[500, 200] = float("nan")
    ref = F.linear(agg, wl, b) + F.linear(x, wr)
    out = _run(agg, x, wl, b, wr, _lib.EPI_NONE).cpu()
    clean = torch.ones(n, dtype=torch.bool)
    clean[17] = clean[500] = False
    assert torch.isfinite(out[clean]).all()
    assert_close(out[clean], ref[clean].double())
    assert torch.isnan(out[500]).all()
    # the split schemes write Inf as Inf + (Inf - Inf): the row comes out NaN where fp32 says +-Inf
    assert not torch.isfinite(out[17]).any() and not torch.isfinite(ref[17]).any()


@pytest.mark.skipif(forced, reason="another kernel family is forced")
def test_more_doubtful_tiles_than_the_list_holds():
    """Every 256-row tile holds a row far below the maximum: 2 x 2344 list entries > 4096, the
    list overflows and the bf16 kernel recomputes every tile."""
    n = 600_000
    s = torch.ones(n)
    s[::200] = 2.0 ** -40
    g = torch.Generator(device="cuda").manual_seed(11)
    agg = torch.randn(n, 256, device="cuda", generator=g) * s.cuda()[:, None]
    x = torch.randn(n, 256, device="cuda", generator=g) * s.cuda()[:, None]
    _, _, wl, b, wr = _case(4, seed=11)
    b.zero_()
    from graphsage_amd.ops import project_fwd_blocked
    out, _ = project_fwd_blocked(agg, x, wl.cuda(), b.cuda(), wr.cuda(), _lib.EPI_NONE, False, False)
    rows = torch.cat([torch.arange(0, n, 200)[:500], torch.arange(1, n, 997)])
    want = _ref64(agg[rows].cpu(), x[rows].cpu(), wl, b, wr, _lib.EPI_NONE)
    _rowwise_close(out[rows], want)


def test_argument_checks():
    from graphsage_amd.ops import project_fwd_blocked
    lib = _lib.load()
    assert lib.gsage_project_fwd_range_supported(8, 256, _lib.EPI_RELU) == 1
    assert lib.gsage_project_fwd_range_supported(256, 256, _lib.EPI_L2NORM) == 1
    assert lib.gsage_project_fwd_range_supported(64, 64, _lib.EPI_RELU) == 1          # (its fp16 two-part kernel)
    assert lib.gsage_project_fwd_range_supported(8, 128, _lib.EPI_L2NORM) == 0
    agg, x, wl, b, wr = [t.cuda() for t in _case(64, f=64, seed=1)]
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):          # 64 -> 64 publishes a maximum only from its fp16 kernel, which wants amax_in
        project_fwd_blocked(agg, x, wl, b, wr, _lib.EPI_RELU, False, False, amax_out=word)
    agg, x, wl, b, wr = [t.cuda() for t in _case(64, seed=1)]
    with pytest.raises(ValueError):          # in and out must be different words
        project_fwd_blocked(agg, x, wl, b, wr, _lib.EPI_RELU, False, False, amax_in=word, amax_out=word)
    with pytest.raises(ValueError):
        project_fwd_blocked(agg, x, wl, b, wr, _lib.EPI_RELU, False, False, amax_in=torch.zeros(1, device="cuda"))
