"""CPU: pins the oracle (oracle/sage_oracle.py) to the hand-computed KATs and to the golden
vectors generated from the reference's own Python (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import kat
from oracle import sage_oracle as O


def T(v, dtype=torch.float32):
    return torch.tensor(v, dtype=dtype)


def EI(v):
    return torch.tensor(v, dtype=torch.int64).reshape(2, -1)


# ------------------------------------------------------------------ SAGEConv KATs
def test_kat1_forward_and_grads():
    k = kat.KAT1
    x = T(k["x"]).requires_grad_()
    wl, bl, wr = T(k["w_l"]).requires_grad_(), T(k["b_l"]).requires_grad_(), T(k["w_r"]).requires_grad_()
    ei = EI(k["edge_index"])
    agg = O.mean_aggregate(x, ei)
    assert torch.equal(agg.detach(), T(k["agg"]))
    out = O.sage_conv(x, ei, wl, bl, wr)
    assert torch.equal(out.detach(), T(k["out"]))
    (out * T(k["G"])).sum().backward()
    assert torch.equal(bl.grad, T(k["grad_b_l"]))
    assert torch.equal(wl.grad, T(k["grad_w_l"]))
    assert torch.equal(wr.grad, T(k["grad_w_r"]))
    assert torch.equal(x.grad, T(k["grad_x"]))


def test_kat1_csr():
    k = kat.KAT1
    rowptr, col, rowptr_t, col_t, inv_deg = O.csr_build(EI(k["edge_index"]), 4)
    assert rowptr.tolist() == k["rowptr"] and col.tolist() == k["col"]
    assert rowptr_t.tolist() == k["rowptr_t"] and col_t.tolist() == k["col_t"]
    assert inv_deg.tolist() == k["inv_deg"]
    assert rowptr.dtype == np.int32 and col.dtype == np.int32 and inv_deg.dtype == np.float32


def test_kat2_no_edges():
    k = kat.KAT2
    out = O.sage_conv(T(k["x"]), EI(k["edge_index"]), T(k["w_l"]), T(k["b_l"]), T(k["w_r"]))
    assert torch.equal(out, T(k["out"]))
    rowptr, col, *_ = O.csr_build(EI(k["edge_index"]), 2)
    assert rowptr.tolist() == k["rowptr"] and col.tolist() == k["col"]


def test_kat3_self_loop():
    k = kat.KAT3
    x, ei = T(k["x"]), EI(k["edge_index"])
    assert torch.equal(O.mean_aggregate(x, ei), T(k["agg"]))
    assert torch.equal(O.sage_conv(x, ei, T(k["w_l"]), T(k["b_l"]), T(k["w_r"])), T(k["out"]))


def test_kat4_normalize():
    k = kat.KAT4
    y = torch.nn.functional.normalize(T(k["x"]), p=2, dim=1)
    assert torch.allclose(y, T(k["y"]), atol=1e-7)


def test_kat5_loss():
    k = kat.KAT5
    loss, det = O.skipgram_loss_with_details(T(k["emb"]), T(k["targets"], torch.int64),
                                             T(k["contexts"], torch.int64),
                                             T(k["negatives"], torch.int64), k["scale"])
    assert abs(loss.item() - k["loss"]) < 1e-6
    assert abs(det["pos_prob"] - k["pos_prob"]) < 1e-6 and abs(det["neg_prob"] - k["neg_prob"]) < 1e-6


# --------------------------------------------------- model structure (graphsage.py:68-157)
def test_model_structure_matches_reference_contract():
    m = O.OracleGraphSAGE(7, 64, 64, 2, 0.3)
    assert m.count_parameters() == 9216            # tests/test_model.py:79-84 bound, exact value
    keys = sorted(m.state_dict().keys())
    assert keys == sorted([f"convs.{i}.{n}" for i in range(2)
                           for n in ("lin_l.weight", "lin_l.bias", "lin_r.weight")])
    assert len(O.OracleGraphSAGE(7, 64, 32, 1).convs) == 1   # num_layers==1 quirk (:103,110)
    assert len(O.OracleGraphSAGE(7, 64, 32, 3).convs) == 3


def test_reference_test_invariants_on_oracle():
    # tests/test_model.py:23-77 of the reference: shapes, unit norms, grads reach every param
    torch.manual_seed(0)
    m = O.OracleGraphSAGE(7, 64, 64, 2, 0.3)
    x, ei = torch.randn(100, 7), torch.randint(0, 100, (2, 500))
    m.eval()
    out = m(x, ei)
    assert out.shape == (100, 64)
    assert torch.allclose(out.norm(dim=1), torch.ones(100), atol=1e-5)
    m.train()
    m(x, ei).sum().backward()
    assert all(p.grad is not None for p in m.parameters())


# ---------------------------------------------------------------- loss goldens (reference)
def _loss_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "loss_golden.npz"))
    return z, [str(n) for n in z["names"]]


def test_loss_oracle_matches_reference_goldens(golden_dir):
    z, names = _loss_cases(golden_dir)
    assert len(names) == 10
    for name in names:
        emb = torch.from_numpy(z[f"{name}.emb"]).clone().requires_grad_()
        t, c, n = (torch.from_numpy(z[f"{name}.{k}"]) for k in ("targets", "contexts", "negatives"))
        scale = float(z[f"{name}.scale"])
        loss, det = O.skipgram_loss_with_details(emb, t, c, n, scale)
        loss.backward()
        # same ops in the same order as loss.py:161-193; reductions may be split differently
        # over CPU threads, so allow a few ulps rather than demanding bitwise equality
        assert loss.item() == pytest.approx(float(z[f"{name}.loss"]), rel=2e-7), name
        ref_det = z[f"{name}.details"]
        for i, k in enumerate(("total_loss", "pos_loss", "neg_loss", "pos_score_mean",
                               "neg_score_mean", "pos_prob", "neg_prob")):
            assert det[k] == pytest.approx(ref_det[i], rel=2e-7, abs=1e-9), (name, k)
        assert np.allclose(emb.grad.numpy(), z[f"{name}.grad"], rtol=1e-6, atol=1e-9), name
        assert O.skipgram_loss(emb.detach(), t, c, n, scale).item() == pytest.approx(
            float(z[f"{name}.loss"]), rel=2e-7)


# ------------------------------------------------------------- feeder goldens (next rows)
def test_alias_table_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "alias_golden.npz"))
    for g in ("g1", "g2", "g3"):
        ei = torch.from_numpy(z[f"{g}.edge_index"])
        probs, prob_table, alias = O.alias_table(ei, int(z[f"{g}.num_nodes"]))
        assert np.array_equal(probs.numpy(), z[f"{g}.probs"]), g
        assert np.array_equal(prob_table.numpy(), z[f"{g}.prob_table"]), g
        assert np.array_equal(alias.numpy(), z[f"{g}.alias"]), g


def test_c1_graph_fixture_contract(golden_dir):
    z = np.load(os.path.join(golden_dir, "c1_graph.npz"))
    ei, x = z["edge_index"], z["features"]
    assert ei.shape == (2, 1106) and ei.dtype == np.int64 and x.shape == (500, 7)
    key = ei[0] * 500 + ei[1]
    assert np.all(np.diff(key) > 0)                           # sorted by (src,dst), no duplicates
    assert set(map(tuple, ei.T)) == set(map(tuple, ei[::-1].T))  # symmetric
    assert x[:, :4].min() >= 0 and x[:, :4].max() <= 1
    assert set(np.unique(x[:, 4:])) <= {0.0, 1.0}
    deg = np.bincount(ei[1], minlength=500)
    assert (deg == 0).sum() == 49 and deg.max() == 7


def test_oracle_snapshot(golden_dir):
    """The oracle's own output on the C1 graph is pinned, so a silent change of the
    restatement (or of torch's CPU kernels) shows up here before it reaches the GPU tests."""
    z = np.load(os.path.join(golden_dir, "c1_graph.npz"))
    s = np.load(os.path.join(golden_dir, "oracle_c1_snapshot.npz"))
    torch.manual_seed(0)
    m = O.OracleGraphSAGE(7, 64, 64, 2, 0.3).eval()
    emb = m(torch.from_numpy(z["features"]), torch.from_numpy(z["edge_index"]))
    assert np.allclose(emb.detach().numpy(), s["embeddings"], rtol=1e-6, atol=1e-7)


def test_two_layer_forward_against_exact_rational_arithmetic():
    """An independent restatement of the spec in exact rational arithmetic,
    plain Python loops, no torch op: mean over in-edges counted with multiplicity (self loops and
    duplicates included, zero row for isolated nodes), lin_l with bias on the mean plus lin_r on
    the node itself, ReLU between the layers.  Inputs are small integers and halves; the means
    bring in thirds and fifths that float32 cannot hold, so the oracle is compared to 1e-6."""
    from fractions import Fraction as Fr
    import random
    rnd = random.Random(7)
    n, f0, f1, f2 = 9, 3, 4, 2
    edges = [(rnd.randrange(n - 1), rnd.randrange(n - 1)) for _ in range(30)]   # node 8 isolated
    edges += [(2, 2), (3, 5), (3, 5), (3, 5)]                                    # self loop, triple edge
    x = [[Fr(rnd.randrange(-4, 5)) for _ in range(f0)] for _ in range(n)]
    def mat(r, c):
        return [[Fr(rnd.randrange(-3, 4), 2) for _ in range(c)] for _ in range(r)]
    wl1, wr1, b1 = mat(f1, f0), mat(f1, f0), [Fr(rnd.randrange(-2, 3)) for _ in range(f1)]
    wl2, wr2, b2 = mat(f2, f1), mat(f2, f1), [Fr(rnd.randrange(-2, 3)) for _ in range(f2)]

    def conv(h, wl, b, wr):
        width = len(h[0])
        out = []
        for i in range(n):
            nb = [s for (s, t) in edges if t == i]
            agg = [sum((h[s][k] for s in nb), Fr(0)) / max(len(nb), 1) for k in range(width)]
            out.append([sum(wl[o][k] * agg[k] for k in range(width)) + b[o] +
                        sum(wr[o][k] * h[i][k] for k in range(width)) for o in range(len(wl))])
        return out

    h1 = [[max(v, Fr(0)) for v in row] for row in conv(x, wl1, b1, wr1)]
    pre = conv(h1, wl2, b2, wr2)

    m = O.OracleGraphSAGE(f0, f1, f2, 2, dropout=0.0)
    tf = lambda a: torch.tensor([[float(v) for v in row] for row in a], dtype=torch.float64)
    with torch.no_grad():
        m.convs[0].lin_l.weight.copy_(tf(wl1)); m.convs[0].lin_l.bias.copy_(tf([b1])[0])
        m.convs[0].lin_r.weight.copy_(tf(wr1))
        m.convs[1].lin_l.weight.copy_(tf(wl2)); m.convs[1].lin_l.bias.copy_(tf([b2])[0])
        m.convs[1].lin_r.weight.copy_(tf(wr2))
    m.eval()
    ei = torch.tensor(edges, dtype=torch.int64).t().contiguous()
    xs = tf(x).to(torch.float32)
    m.normalize_output = False
    got_pre = m(xs, ei).to(torch.float64)
    want_pre = tf(pre)
    # float32 cannot hold thirds and fifths exactly: 1e-6 relative on the un-normalised output
    assert torch.allclose(got_pre, want_pre, rtol=1e-6, atol=1e-6)
    m.normalize_output = True
    got = m(xs, ei).to(torch.float64)
    norms = want_pre.norm(dim=1, keepdim=True).clamp(min=1e-12)
    assert torch.allclose(got, want_pre / norms, rtol=1e-6, atol=1e-6)
    assert torch.equal(got[8] != 0, (want_pre[8] != 0))      # the isolated node still gets b_l + W_r x


def test_oracle_layer_equals_the_sparse_matrix_form_in_scipy():
    """A second, independent derivation of the SAGEConv arithmetic (PyG's documented formula
    x_i' = W_r x_i + W_l mean_{j in N(i)} x_j + b): the mean as a row-normalised sparse adjacency
    product D^-1 A x built with scipy.sparse (coo -> csr sums duplicate edges, so multiplicities
    count; a row without in-edges stays zero), forward and parameter gradients in float64 against
    the oracle's index_select / scatter_add restatement.  The reference holds no numbers for this
    layer (parity unpinned); this pins the two derivations to each other."""
    import numpy as np
    import scipy.sparse as sp
    g = torch.Generator().manual_seed(5)
    n, e, fi, fo = 300, 2500, 7, 16
    ei = torch.randint(0, n, (2, e), generator=g)          # duplicates and self-loops occur
    ei[1, :40] = 7                                          # a hub target
    ei = ei[:, ei[1] != 11]                                 # node 11 has no in-edges
    x = torch.randn(n, fi, generator=g, dtype=torch.float64)
    conv = O.OracleSAGEConv(fi, fo).double()
    out = conv(x.clone().requires_grad_(False), ei)
    A = sp.coo_matrix((np.ones(ei.size(1)), (ei[1].numpy(), ei[0].numpy())), shape=(n, n)).tocsr()
    deg = np.maximum(np.asarray(A.sum(1)).ravel(), 1.0)
    agg = sp.diags(1.0 / deg) @ A @ x.numpy()
    assert np.abs(agg[11]).max() == 0.0
    w_l, b_l, w_r = (conv.lin_l.weight.detach().numpy(), conv.lin_l.bias.detach().numpy(),
                     conv.lin_r.weight.detach().numpy())
    want = agg @ w_l.T + b_l + x.numpy() @ w_r.T
    assert np.abs(out.detach().numpy() - want).max() <= 1e-12
    # gradients of sum(out * P): dW_l = P^T agg, dW_r = P^T x, db = colsum P, dx = W_r^T P + A^T D^-1 P W_l
    P = torch.randn(n, fo, generator=g, dtype=torch.float64)
    xg = x.clone().requires_grad_(True)
    (conv(xg, ei) * P).sum().backward()
    p = P.numpy()
    assert np.abs(conv.lin_l.weight.grad.numpy() - p.T @ agg).max() <= 1e-10
    assert np.abs(conv.lin_r.weight.grad.numpy() - p.T @ x.numpy()).max() <= 1e-10
    assert np.abs(conv.lin_l.bias.grad.numpy() - p.sum(0)).max() <= 1e-10
    dx = p @ w_r + (A.T @ (sp.diags(1.0 / deg) @ (p @ w_l)))
    assert np.abs(xg.grad.numpy() - dx).max() <= 1e-10
