"""Comparison helper for the parity tests: 1e-5 relative (north_star), with an absolute floor
of 2e-6 x max|want| (elementwise relative error is undefined at zeros)."""
import torch


def assert_close(got, want, rtol=1e-5, floor=2e-6, what=""):
    got = got.detach().cpu().to(torch.float64)
    want = want.detach().cpu().to(torch.float64)
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = float(want.abs().max()) if want.numel() else 1.0
    scale = scale if scale > 0 else 1.0
    err = (got - want).abs()
    tol = rtol * want.abs() + floor * scale
    bad = err > tol
    if bool(bad.any()):
        i = int(torch.argmax(err - tol))
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} elements off; worst |err|={float(err.flatten()[i]):.3e} "
            f"at flat index {i} (got {float(got.flatten()[i]):.8g}, want {float(want.flatten()[i]):.8g}, "
            f"tol {float(tol.flatten()[i]):.3e}, scale {scale:.3g})")
