"""GPU: K1 CSR build, bit-exact against the oracle (bincount + stable argsort)."""
import numpy as np
import pytest
import torch

from oracle import kat
from oracle import sage_oracle as O

pytestmark = pytest.mark.gpu


def _check(ei_cpu, n):
    from graphsage_amd.graph import build_csr
    g = build_csr(ei_cpu.cuda(), n)
    rowptr, col, rowptr_t, col_t, inv_deg = O.csr_build(ei_cpu, n)
    assert np.array_equal(g.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(g.col.cpu().numpy(), col)
    assert np.array_equal(g.rowptr_t.cpu().numpy(), rowptr_t)
    assert np.array_equal(g.col_t.cpu().numpy(), col_t)
    assert np.array_equal(g.inv_deg.cpu().numpy(), inv_deg)
    assert g.rowptr.dtype == torch.int32 and g.col.dtype == torch.int32
    return g


def test_kat1():
    k = kat.KAT1
    g = _check(torch.tensor(k["edge_index"], dtype=torch.int64), 4)
    assert g.rowptr.tolist() == k["rowptr"] and g.col.tolist() == k["col"]
    assert g.col_t.tolist() == k["col_t"] and g.inv_deg.tolist() == k["inv_deg"]


def test_empty_edge_list():
    g = _check(torch.zeros(2, 0, dtype=torch.int64), 5)
    assert g.rowptr.tolist() == [0] * 6 and g.inv_deg.tolist() == [1.0] * 5


@pytest.mark.parametrize("n,e,seed", [
    (100, 500, 0),            # reference test shape (tests/test_model.py:41): dups + self loops
    (1, 10, 1),               # one node, all self loops
    (2, 1, 2),
    (255, 4000, 3),           # 1 radix pass
    (256, 5000, 4),           # key N needs 9 bits -> 2 passes
    (5000, 4096, 5),          # exactly one tile
    (5000, 4097, 6),          # tile + 1
    (70000, 300000, 7),       # 17 bits -> 3 passes
    (1 << 20, 3000000, 8),    # 21 bits, many tiles
])
def test_random_multigraph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    _check(torch.randint(0, n, (2, e), generator=g), n)


def test_c1_fixture_sorted_symmetric(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "c1_graph.npz"))
    g = _check(torch.from_numpy(z["edge_index"]), 500)
    # symmetric + sorted input: both orientations coincide
    assert torch.equal(g.rowptr, g.rowptr_t) and torch.equal(g.col, g.col_t)


def test_hub_rows_listed():
    from graphsage_amd import _lib
    thr = _lib.load().gsage_long_row_threshold()
    n = 3000
    hub_in = torch.stack([torch.randint(0, n, (thr + 500,)), torch.full((thr + 500,), 7)])
    hub_out = torch.stack([torch.full((thr + 1,), 11), torch.randint(0, n, (thr + 1,))])
    rest = torch.randint(0, n, (2, 20000))
    ei = torch.cat([hub_in, rest, hub_out], dim=1)[:, torch.randperm(hub_in.size(1) + 20000 + hub_out.size(1))]
    g = _check(ei, n)
    nl = g.n_long.tolist()
    deg_in, deg_out = g.in_degree(), g.out_degree()
    assert sorted(g.long_rows[:nl[0]].tolist()) == sorted(torch.nonzero(deg_in > thr).flatten().tolist())
    assert sorted(g.long_rows_t[:nl[1]].tolist()) == sorted(torch.nonzero(deg_out > thr).flatten().tolist())
    assert 7 in g.long_rows[:nl[0]].tolist() and 11 in g.long_rows_t[:nl[1]].tolist()


def test_out_of_range_index_raises():
    from graphsage_amd.graph import build_csr
    ei = torch.tensor([[0, 1, 9], [1, 2, 0]], dtype=torch.int64).cuda()
    with pytest.raises(IndexError):
        build_csr(ei, 3)
    ei = torch.tensor([[0, 1, 2], [1, -1, 0]], dtype=torch.int64).cuda()
    with pytest.raises(IndexError):
        build_csr(ei, 3)


def test_rejects_wrong_inputs():
    from graphsage_amd.graph import build_csr
    with pytest.raises(ValueError):
        build_csr(torch.zeros(3, 4, dtype=torch.int64).cuda(), 4)
    with pytest.raises(ValueError):
        build_csr(torch.zeros(2, 4, dtype=torch.int32).cuda(), 4)
    with pytest.raises(ValueError):
        build_csr(torch.zeros(2, 4, dtype=torch.int64), 4)   # CPU tensor: no CPU path


def test_deterministic_and_full_size_properties():
    """C2-sized power-law graph: properties that need no oracle at full size."""
    from graphsage_amd.graph import build_csr
    from graphsage_amd.synth import powerlaw_graph
    n = 100_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    a, b = build_csr(ei, n), build_csr(ei, n)
    for f in ("rowptr", "col", "rowptr_t", "col_t", "inv_deg"):
        assert torch.equal(getattr(a, f), getattr(b, f))     # run-to-run identical
    E = ei.size(1)
    assert int(a.rowptr[-1]) == E and int(a.rowptr_t[-1]) == E
    assert torch.equal(a.in_degree(), torch.bincount(ei[1], minlength=n))
    # symmetric sorted input -> col_t is exactly ei[1] and CSR == CSR^T
    assert torch.equal(a.col_t.to(torch.int64), ei[1])
    assert torch.equal(a.col, a.col_t)
    _check(ei.cpu(), n)
