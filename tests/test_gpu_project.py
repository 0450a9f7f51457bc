"""GPU: K3 fused projection forward/backward vs plain torch fp32 (F.linear / relu /
dropout-with-given-mask / F.normalize) on the CPU.  MFMA and VALU paths both."""
import pytest
import torch
import torch.nn.functional as F

from graphsage_amd import _lib

from util import assert_close

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 2e-6

EPIS = [_lib.EPI_NONE, _lib.EPI_RELU, _lib.EPI_RELU_DROPOUT, _lib.EPI_L2NORM]


def _ref(agg, x, wl, bl, wr, epi, keep, p):
    pre = F.linear(agg, wl, bl) + F.linear(x, wr)
    if epi == _lib.EPI_NONE:
        return pre
    if epi == _lib.EPI_RELU:
        return F.relu(pre)
    if epi == _lib.EPI_RELU_DROPOUT:
        return F.relu(pre) * keep.to(torch.float32) / (1.0 - p)
    return F.normalize(pre, p=2, dim=1)


def _case(n, fi, fo, seed):
    g = torch.Generator().manual_seed(seed)
    agg = torch.randn(n, fi, generator=g)
    x = torch.randn(n, fi, generator=g)
    wl = torch.randn(fo, fi, generator=g) / fi ** 0.5
    wr = torch.randn(fo, fi, generator=g) / fi ** 0.5
    bl = torch.randn(fo, generator=g)
    keep = (torch.rand(n, fo, generator=g) >= 0.3).to(torch.uint8)
    gout = torch.randn(n, fo, generator=g)
    return agg, x, wl, bl, wr, keep, gout


SHAPES = [(7, 64), (64, 64), (64, 128), (128, 128), (128, 64), (256, 256), (7, 32), (32, 32),
          (5, 3), (64, 96), (20, 64), (7, 100), (8, 128), (7, 256), (8, 64)]


# force_generic 0 = the kernel the library picks for the shape; 1 = the VALU kernels, tried where
# 0 would pick something else; 2 = the generic MFMA tiles in place of the 64 -> 64 special case
CASES = [(fi, fo, 0) for fi, fo in SHAPES] + \
        [(fi, fo, 1) for fi, fo in [(64, 64), (128, 64), (7, 64)]] + [(64, 64, 2)]


@pytest.mark.parametrize("epi", EPIS)
@pytest.mark.parametrize("fi,fo,force_generic", CASES)
def test_forward_backward(fi, fo, epi, force_generic):
    from graphsage_amd.ops import project_bwd, project_fwd
    n, p = 333, 0.3          # not a multiple of the 64-row tile
    agg, x, wl, bl, wr, keep, gout = _case(n, fi, fo, fi * 1000 + fo)
    leaves = [t.clone().requires_grad_() for t in (agg, x, wl, bl, wr)]
    ref = _ref(*leaves, epi, keep, p)
    ref.backward(gout)
    d = [t.cuda() for t in (agg, x, wl, bl, wr)]
    out, rnorm = project_fwd(*d, epilogue=epi, keep_mask=keep.cuda() if epi == _lib.EPI_RELU_DROPOUT else None,
                             p=p if epi == _lib.EPI_RELU_DROPOUT else 0.0, force_generic=force_generic)
    assert_close(out.cpu(), ref.detach())
    scale = torch.rand(n).cuda() + 0.5
    ga, gx, gwl, gb, gwr = project_bwd(gout.cuda(), out, rnorm, d[0], d[1], d[2], d[4], scale, epi,
                                       p if epi == _lib.EPI_RELU_DROPOUT else 0.0, True,
                                       force_generic=force_generic)
    assert_close((ga / scale[:, None]).cpu(), leaves[0].grad)
    assert_close(gx.cpu(), leaves[1].grad)
    assert_close(gwl.cpu(), leaves[2].grad)
    assert_close(gb.cpu(), leaves[3].grad)
    assert_close(gwr.cpu(), leaves[4].grad)


def test_mfma_identity_asymmetric():
    """A = I against an asymmetric B: catches a transposed C write or a swapped operand map."""
    from graphsage_amd.ops import project_fwd
    n = 64
    agg = torch.eye(64)[:n].contiguous()
    x = torch.zeros(n, 64)
    wl = (torch.arange(64 * 64, dtype=torch.float32).reshape(64, 64) % 97) - 40.0   # wl[j,k] != wl[k,j]
    wr = torch.zeros(64, 64)
    bl = torch.zeros(64)
    out, _ = project_fwd(agg.cuda(), x.cuda(), wl.cuda(), bl.cuda(), wr.cuda())
    assert torch.equal(out.cpu(), wl.t().contiguous())     # out[i,j] = wl[j,i], exact integers


def test_l2norm_zero_row_and_rnorm():
    from graphsage_amd.ops import project_bwd, project_fwd
    n, f = 70, 64
    agg = torch.zeros(n, f)
    x = torch.zeros(n, f)
    x[1:] = torch.randn(n - 1, f)
    wl, wr, bl = torch.randn(f, f), torch.randn(f, f), torch.zeros(f)   # row 0 -> pre = 0
    leaves = [t.clone().requires_grad_() for t in (agg, x, wl, bl, wr)]
    ref = F.normalize(F.linear(leaves[0], leaves[2], leaves[3]) + F.linear(leaves[1], leaves[4]), dim=1)
    out, rnorm = project_fwd(agg.cuda(), x.cuda(), wl.cuda(), bl.cuda(), wr.cuda(), epilogue=_lib.EPI_L2NORM)
    assert torch.count_nonzero(out[0]).item() == 0          # 0 / max(0, 1e-12) = 0 (KAT-4)
    assert_close(out.cpu(), ref.detach())
    assert torch.allclose(out[1:].norm(dim=1).cpu(), torch.ones(n - 1), atol=1e-5)


def test_inkernel_dropout_matches_host_mirror():
    from graphsage_amd.ops import project_fwd
    from graphsage_amd.rng import dropout_keep_mask
    p, seed = 0.3, 0x1234_5678_9ABC_DEF1
    for (n, fi, fo, fg) in [(500, 7, 64, False), (200, 64, 64, False), (200, 64, 128, False), (100, 20, 48, False),
                           (300, 256, 256, False), (130, 128, 128, False)]:
        agg, x, wl, bl, wr, _, _ = _case(n, fi, fo, 5)
        d = [t.cuda() for t in (agg, x, wl, bl, wr)]
        out, _ = project_fwd(*d, epilogue=_lib.EPI_RELU_DROPOUT, seed=seed, p=p, force_generic=fg)
        keep = torch.from_numpy(dropout_keep_mask(seed, n, fo, p))
        ref = _ref(agg, x, wl, bl, wr, _lib.EPI_RELU_DROPOUT, keep, p)
        assert_close(out.cpu(), ref)
        frac = keep.float().mean().item()
        assert abs(frac - (1 - p)) < 0.02


@pytest.mark.parametrize("extra", [0, 512])
def test_inkernel_dropout_at_the_32_bit_element_index_boundary(extra):
    """The input layer's kernel draws its masks four at a time on 32-bit element indices while every index of
    the launch fits (N x Fo <= 2^32) and on 64-bit ones beyond: 2^24 rows x 256 columns is the last launch of
    the first kind, 512 rows more the first of the second.  Rows at both ends of the table against
    relu(pre) x mask / (1 - p) with the host mirror's mask at the same 64-bit indices (graphsage.py:134-135)."""
    from graphsage_amd.ops import project_fwd
    from graphsage_amd.rng import dropout_keep_mask_range
    n, fo, p, seed = (1 << 24) + extra, 256, 0.3, 0xFEDC_BA98_7654_3211
    g = torch.Generator(device="cuda").manual_seed(3)
    agg = torch.rand(n, 8, device="cuda", generator=g)
    x = torch.rand(n, 8, device="cuda", generator=g)
    agg[:, 7] = 0
    x[:, 7] = 0
    w_l = torch.randn(fo, 8, device="cuda", generator=g) / 3
    w_r = torch.randn(fo, 8, device="cuda", generator=g) / 3
    b = torch.randn(fo, device="cuda", generator=g)
    out, _ = project_fwd(agg, x, w_l, b, w_r, epilogue=_lib.EPI_RELU_DROPOUT, seed=seed, p=p)
    for first, count in ((0, 300), ((1 << 23) - 100, 200), (n - 700, 700)):
        rows = slice(first, first + count)
        pre = agg[rows].double() @ w_l.double().T + b.double() + x[rows].double() @ w_r.double().T
        keep = torch.from_numpy(dropout_keep_mask_range(seed, first * fo, count * fo, p)).view(count, fo)
        want = torch.relu(pre).cpu() * keep.double() / (1 - p)
        assert_close(out[rows].cpu(), want, what=f"rows {first}..{first + count}")
    del out


def test_weight_grads_bitwise_reproducible():
    from graphsage_amd.ops import project_bwd, project_fwd
    n, fi, fo = 5000, 64, 64
    agg, x, wl, bl, wr, keep, gout = _case(n, fi, fo, 9)
    d = [t.cuda() for t in (agg, x, wl, bl, wr)]
    out, rn = project_fwd(*d, epilogue=_lib.EPI_L2NORM)
    r1 = project_bwd(gout.cuda(), out, rn, d[0], d[1], d[2], d[4], None, _lib.EPI_L2NORM, 0.0, True)
    r2 = project_bwd(gout.cuda(), out, rn, d[0], d[1], d[2], d[4], None, _lib.EPI_L2NORM, 0.0, True)
    for a, b in zip(r1, r2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("epi", EPIS)
@pytest.mark.parametrize("n", [333, 32, 4097])
def test_backward_zero_row_mask(epi, n):
    """gsage_project_bwd_nz: a clear bit only where the row of grad_agg / grad_x is all zero, and
    (for the fused 64 -> 64 kernel) a set bit exactly where the pre-epilogue gradient is not."""
    from graphsage_amd.ops import project_bwd, project_fwd
    agg, x, wl, bl, wr, keep, gout = _case(n, 64, 64, n)
    zero_rows = torch.rand(n, generator=torch.Generator().manual_seed(n)) < 0.6
    gout[zero_rows] = 0.0
    d = [t.cuda() for t in (agg, x, wl, bl, wr)]
    p = 0.3 if epi == _lib.EPI_RELU_DROPOUT else 0.0
    out, rnorm = project_fwd(*d, epilogue=epi, keep_mask=keep.cuda() if p else None, p=p)
    ga, gx, gwl, gb, gwr, nz = project_bwd(gout.cuda(), out, rnorm, d[0], d[1], d[2], d[4], None,
                                           epi, p, True, want_nz_mask=True)
    assert nz is not None and nz.dtype == torch.int32 and nz.numel() == (n + 31) // 32
    bits = ((nz.to(torch.int64)[:, None] >> torch.arange(32, device="cuda")) & 1).reshape(-1)
    assert int(bits[n:].sum()) == 0                      # rows past N are reported as zero
    bits = bits[:n].bool()
    nonzero = (ga != 0).any(1) | (gx != 0).any(1)
    assert not bool((nonzero & ~bits).any())             # never hides a non-zero row
    assert not bool((bits & zero_rows.cuda()).any())     # and finds every zero gradient row
    # same numbers as the call without the mask
    ga0, gx0, gwl0, gb0, gwr0 = project_bwd(gout.cuda(), out, rnorm, d[0], d[1], d[2], d[4], None,
                                            epi, p, True)
    for a, b in ((ga, ga0), (gx, gx0), (gwl, gwl0), (gb, gb0), (gwr, gwr0)):
        assert torch.equal(a, b)
    # the generic kernels do not look: every row is reported as possibly non-zero
    nz2 = project_bwd(gout.cuda(), out, rnorm, d[0], d[1], d[2], d[4], None, epi, p, True,
                      force_generic=2, want_nz_mask=True)[5]
    assert bool((nz2 == -1).all())
    # other shapes have no kernel that looks: no mask is handed out
    a2, x2, wl2, bl2, wr2, _, g2 = _case(n, 128, 64, 1)
    d2 = [t.cuda() for t in (a2, x2, wl2, bl2, wr2)]
    o2, r2 = project_fwd(*d2, epilogue=_lib.EPI_NONE)
    assert project_bwd(g2.cuda(), o2, r2, d2[0], d2[1], d2[2], d2[4], None, _lib.EPI_NONE, 0.0, True,
                       want_nz_mask=True)[5] is None


def test_wide_shapes_on_both_kernel_families():
    """The 128/256-wide layers have three kernel families (fp32 MFMA, bf16 MFMA on the three-way
    split of every fp32 operand, fp16 MFMA on the range-scaled two-way split) chosen by shape;
    GSAGE_WIDE_GEMM forces one -- "half" also onto the 128-column tiles and the dX GEMMs it is not
    chosen for.  The library reads it once, so each family runs this file's forward/backward
    cases in a child process."""
    import os
    import subprocess
    import sys
    if os.environ.get("GSAGE_WIDE_GEMM"):
        pytest.skip("already inside the forced run")
    here = os.path.dirname(os.path.abspath(__file__))
    for mode in ("split", "f32", "half"):
        env = dict(os.environ, GSAGE_WIDE_GEMM=mode)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                            os.path.join(here, "test_gpu_project.py"), "-k",
                            "test_forward_backward and (128 or 256)"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"GSAGE_WIDE_GEMM={mode}:\n{r.stdout[-3000:]}\n{r.stderr[-1000:]}"
        assert " passed" in r.stdout


@pytest.mark.parametrize("epi", [_lib.EPI_RELU, _lib.EPI_RELU_DROPOUT])
@pytest.mark.parametrize("n,fi,fo", [(333, 8, 64), (4099, 8, 128), (64, 8, 256), (333, 5, 64), (1, 8, 64)])
def test_relu_sign_bits_replace_the_saved_output(n, fi, fo, epi):
    """gsage_project_fwd_ex / _bwd_ex: the first-layer kernel leaves `out > 0` bits, the dW kernel
    uses them instead of the saved output and gives bit-identical weight gradients; forward
    kernels that leave no bits (Fi != 8 here) say so and the backward reads `out` as before."""
    from graphsage_amd.ops import project_bwd, project_fwd
    agg, x, wl, bl, wr, keep, gout = _case(n, fi, fo, 7 * n + fo)
    d = [t.cuda() for t in (agg, x, wl, bl, wr)]
    p = 0.3 if epi == _lib.EPI_RELU_DROPOUT else 0.0
    km = keep.cuda() if p else None
    out, rnorm, sign = project_fwd(*d, epilogue=epi, keep_mask=km, p=p, want_relu_sign=True)
    out0, _ = project_fwd(*d, epilogue=epi, keep_mask=km, p=p)
    assert torch.equal(out, out0)
    assert sign is not None and sign.dtype == torch.int64
    ng, ncb = (n + 3) // 4, fo // 64
    flag = int(sign[-1].item())
    assert flag == (1 if fi == 8 else 0)
    if flag:
        # word c of (column block, 4-row group): bit l = (row 4g + l // 16, column 64cb + 4 (l % 16) + c)
        w = sign[:-1].view(ncb, ng, 4)
        bits = ((w[..., None] >> torch.arange(64, device="cuda")) & 1).bool()     # [cb, g, c, l]
        pos = torch.zeros(ng * 4, fo, dtype=torch.bool, device="cuda")
        pos[:n] = out > 0
        want = pos.view(ng, 4, ncb, 16, 4).permute(2, 0, 4, 1, 3).reshape(ncb, ng, 4, 64)
        live = (torch.arange(ng * 4, device="cuda") < n).view(ng, 4)[None, :, None, :, None]
        live = live.expand(ncb, ng, 4, 4, 16).reshape(ncb, ng, 4, 64)
        assert torch.equal(bits & live, want & live)
    g = gout.cuda()
    ref = project_bwd(g, out, rnorm, d[0], d[1], d[2], d[4], None, epi, p, False)
    got = project_bwd(g, out, rnorm, d[0], d[1], d[2], d[4], None, epi, p, False, relu_sign=sign)
    for a, b in zip(ref[2:], got[2:]):
        assert torch.equal(a, b)
    # the bits are what is read: a scrambled `out` must not change the weight gradients
    if flag:
        got2 = project_bwd(g, torch.full_like(out, -1.0), rnorm, d[0], d[1], d[2], d[4], None, epi, p,
                           False, relu_sign=sign)
        for a, b in zip(ref[2:], got2[2:]):
            assert torch.equal(a, b)


def _float_bits(v: float) -> torch.Tensor:
    return torch.tensor([v], dtype=torch.float32).view(torch.int32).cuda()


@pytest.mark.parametrize("epi", EPIS)
@pytest.mark.parametrize("n", [1, 333, 4097])
def test_64x64_forward_on_the_fp16_pipe_when_the_input_range_is_given(n, epi):
    """A 64 -> 64 layer that is told max |x| (amax_in) runs proj64h_fwd_k: two-part fp16 operands,
    three products.  Same bar as the fp32 kernel, against fp64; amax_out = max |out| exactly."""
    from graphsage_amd.ops import project_fwd
    p = 0.3
    agg, x, wl, bl, wr, keep, _ = _case(n, 64, 64, 77 + n)
    agg = agg.clamp(-x.abs().max(), x.abs().max())          # a mean of rows of x stays inside its range
    ref = _ref(*(t.double() for t in (agg, x, wl, bl, wr)), epi, keep, p).float()
    d = [t.cuda() for t in (agg, x, wl, bl, wr)]
    a_in, a_out = _float_bits(x.abs().max().item()), torch.full((1,), -1, dtype=torch.int32).cuda()
    kw = dict(epilogue=epi, keep_mask=keep.cuda() if epi == _lib.EPI_RELU_DROPOUT else None,
              p=p if epi == _lib.EPI_RELU_DROPOUT else 0.0)
    out, _ = project_fwd(*d, amax_in=a_in, amax_out=a_out, **kw)
    assert_close(out.cpu(), ref)
    assert a_out.view(torch.float32).item() == out.abs().max().item()
    plain, _ = project_fwd(*d, **kw)                        # the fp32 kernel on the same operands
    assert_close(out.cpu(), plain.cpu())


def test_64x64_fp16_pipe_wide_dynamic_range_and_zero_input():
    """Operands spanning eight decades under one bound: the error stays at the scale of the largest
    pre-activation (the documented bound).  All-zero input (bound 0): bias only, no NaN."""
    from graphsage_amd.ops import project_fwd
    n = 500
    agg, x, wl, bl, wr, _, _ = _case(n, 64, 64, 5)
    mag = 10.0 ** torch.randint(-6, 3, (n, 1)).float()
    x, agg = x * mag, agg * mag
    bound = max(x.abs().max().item(), agg.abs().max().item())
    ref = _ref(*(t.double() for t in (agg, x, wl, bl, wr)), _lib.EPI_NONE, None, 0.0)
    d = [t.cuda() for t in (agg, x, wl, bl, wr)]
    out, _ = project_fwd(*d, amax_in=_float_bits(bound))
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * ref.abs().max().item(), err
    z = torch.zeros(n, 64).cuda()
    out0, _ = project_fwd(z, z, d[2], d[3], d[4], amax_in=_float_bits(0.0))
    assert torch.equal(out0.cpu(), bl.expand(n, 64))
