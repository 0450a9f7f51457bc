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
