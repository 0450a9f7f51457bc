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
