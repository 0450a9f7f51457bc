"""GPU: K2 / K7 segment gather vs the oracle's index_select + scatter_add_ mean."""
import pytest
import torch

from oracle import kat
from oracle import sage_oracle as O

from util import assert_close

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-5, 1e-6   # north_star: 1e-5 relative (fp32)


def _graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (2, e), generator=g)


def test_kat1_exact():
    from graphsage_amd.graph import build_csr
    from graphsage_amd.ops import aggregate_mean
    k = kat.KAT1
    g = build_csr(torch.tensor(k["edge_index"], dtype=torch.int64).cuda(), 4)
    agg = aggregate_mean(g, torch.tensor(k["x"]).cuda())
    assert torch.equal(agg.cpu(), torch.tensor(k["agg"]))


@pytest.mark.parametrize("f", [7, 8, 4, 64, 128, 256, 32, 16, 1, 5, 100, 300])
def test_forward_matches_oracle(f):
    from graphsage_amd.graph import build_csr
    from graphsage_amd.ops import aggregate_mean
    n, e = 1000, 12000
    ei = _graph(n, e, f)
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(f + 1))
    ref = O.mean_aggregate(x, ei)
    got = aggregate_mean(build_csr(ei.cuda(), n), x.cuda())
    assert_close(got.cpu(), ref)


@pytest.mark.parametrize("f", [7, 64, 128])
def test_backward_matches_autograd_of_oracle(f):
    from graphsage_amd.graph import build_csr
    from graphsage_amd.ops import aggregate_mean_bwd
    n, e = 800, 9000
    ei = _graph(n, e, 10 + f)
    x = torch.randn(n, f).requires_grad_()
    gout = torch.randn(n, f)
    O.mean_aggregate(x, ei).backward(gout)
    g = build_csr(ei.cuda(), n)
    got = aggregate_mean_bwd(g, gout.cuda())
    assert_close(got.cpu(), x.grad)
    # prescaled form + accumulate into an existing gradient
    base = torch.randn(n, f)
    got2 = aggregate_mean_bwd(g, (gout.cuda() * g.inv_deg[:, None]), prescaled=True,
                              add_to=base.cuda().clone())
    assert_close(got2.cpu(), x.grad + base)


@pytest.mark.parametrize("f", [64, 7, 8, 256])
def test_hub_rows_and_isolated_rows(f):
    from graphsage_amd import _lib
    from graphsage_amd.graph import build_csr
    from graphsage_amd.ops import aggregate_mean, aggregate_mean_bwd
    thr = _lib.load().gsage_long_row_threshold()
    n = 4000
    hub = torch.stack([torch.randint(0, n, (3 * thr + 77,)), torch.full((3 * thr + 77,), 5)])
    ei = torch.cat([hub, _graph(n - 100, 30000, 3)], dim=1)     # last 100 nodes isolated
    x = torch.randn(n, f).requires_grad_()
    ref = O.mean_aggregate(x, ei)
    g = build_csr(ei.cuda(), n)
    assert g.n_long[0].item() >= 1
    got = aggregate_mean(g, x.detach().cuda())
    assert_close(got.cpu(), ref.detach())
    assert torch.count_nonzero(got[-100:]).item() == 0
    gout = torch.randn(n, f)
    ref.backward(gout)
    gb = aggregate_mean_bwd(g, gout.cuda())
    assert_close(gb.cpu(), x.grad)


def test_bitwise_reproducible_and_linear_at_c2_size():
    """Full-size properties on the C2 graph: determinism and linearity of the operator."""
    from graphsage_amd.graph import build_csr
    from graphsage_amd.ops import aggregate_mean
    from graphsage_amd.synth import powerlaw_graph
    n = 100_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    g = build_csr(ei, n)
    x = torch.randn(n, 64, device="cuda")
    y = torch.randn(n, 64, device="cuda")
    a1, a2 = aggregate_mean(g, x), aggregate_mean(g, x)
    assert torch.equal(a1, a2)
    lin = aggregate_mean(g, 2.0 * x + y)
    assert_close(lin, 2.0 * a1 + aggregate_mean(g, y), rtol=1e-4, floor=1e-5)
    ones = aggregate_mean(g, torch.ones(n, 64, device="cuda"))
    has_in = (g.in_degree() > 0).to(torch.float32)[:, None].expand(-1, 64)
    assert_close(ones, has_in)   # mean of ones is 1 (0 if isolated)
    ref = O.mean_aggregate(x.cpu(), ei.cpu())
    assert_close(a1.cpu(), ref)



def test_full_size_c4_conservation_identities():
    """BASELINE size (10M nodes / ~200M edges): checks that need no oracle.
      sum_i deg_in(i) * mean_agg(x)[i]  ==  sum_j deg_out(j) * x[j]     (every edge counted once)
      transposed sum-gather of ones     ==  out-degree
    and the CSR invariants rowptr[N] == E, rowptr monotone, col within range."""
    from graphsage_amd.graph import build_csr
    from graphsage_amd.ops import aggregate_mean, aggregate_mean_bwd
    from graphsage_amd.synth import powerlaw_graph
    n = 10_000_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    E = ei.size(1)
    g = build_csr(ei, n)
    assert int(g.rowptr[-1]) == E and int(g.rowptr_t[-1]) == E
    assert bool((g.rowptr[1:] >= g.rowptr[:-1]).all())
    assert int(g.col.min()) >= 0 and int(g.col.max()) < n
    assert torch.equal(g.col_t.to(torch.int64), ei[1])            # input is sorted by source
    x = torch.rand(n, 64, device="cuda")
    agg = aggregate_mean(g, x)
    lhs = (agg.double() * g.in_degree()[:, None]).sum(0)
    rhs = (x.double() * g.out_degree()[:, None]).sum(0)
    assert torch.allclose(lhs, rhs, rtol=1e-6)
    ones = torch.ones(n, 64, device="cuda")
    outdeg = aggregate_mean_bwd(g, ones, prescaled=True)
    assert torch.equal(outdeg[:, 0].to(torch.int64), g.out_degree())
    assert torch.equal(aggregate_mean(g, x), agg)                  # bitwise reproducible
