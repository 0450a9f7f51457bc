"""Hand-computed known-answer tests for the SAGEConv arithmetic.  TEST
INFRASTRUCTURE ONLY (see oracle/sage_oracle.py header).

PyG (home of ``SAGEConv``, reference ``src/model/graphsage.py:30``) cannot be
imported here and the reference's tests pin no numbers for it
(``tests/test_model.py:45-99``), so these integer-valued cases, worked by hand
from the published semantics, pin the oracle; the oracle then
pins the HIP kernels.  Every expected value below is an exact small integer or
dyadic rational, so comparisons are exact in fp32.
"""
import math

# ---- KAT-1: 4 nodes, duplicate edge 2->0, node 3 isolated, F_in=2, F_out=3
KAT1 = dict(
    x=[[1., 2.], [3., 4.], [5., 6.], [7., 8.]],
    edge_index=[[0, 0, 1, 2, 2], [1, 2, 2, 0, 0]],
    w_l=[[1., 0.], [0., 1.], [1., 1.]],
    b_l=[1., -1., 0.],
    w_r=[[2., 0.], [0., -1.], [1., -1.]],
    # in-neighbour lists: 0:[2,2] 1:[0] 2:[0,1] 3:[]
    rowptr=[0, 2, 3, 5, 5], col=[2, 2, 0, 0, 1],
    rowptr_t=[0, 2, 3, 5, 5], col_t=[1, 2, 2, 0, 0],
    inv_deg=[0.5, 1.0, 0.5, 1.0],
    agg=[[5., 6.], [1., 2.], [2., 3.], [0., 0.]],
    out=[[8., 3., 10.], [8., -3., 2.], [13., -4., 4.], [15., -9., -1.]],
    # L = sum(out * G)
    G=[[1., 0., 2.], [0., 1., 0.], [1., 1., 1.], [2., 0., -1.]],
    grad_b_l=[4., 2., 2.],
    grad_w_l=[[7., 9.], [3., 5.], [12., 15.]],
    grad_w_r=[[20., 24.], [8., 10.], [0., 2.]],
    grad_agg=[[3., 2.], [0., 1.], [2., 2.], [1., -1.]],
    grad_x=[[5., 0.], [1., 0.], [6., 0.], [3., 1.]],
)

# ---- KAT-2: no edges at all -> out = b_l + x W_r^T
KAT2 = dict(
    x=[[1., 2.], [3., 4.]],
    edge_index=[[], []],
    w_l=[[1., 0.], [0., 1.], [1., 1.]],
    b_l=[1., -1., 0.],
    w_r=[[2., 0.], [0., -1.], [1., -1.]],
    rowptr=[0, 0, 0], col=[],
    out=[[3., -3., -1.], [7., -5., -1.]],
)

# ---- KAT-3: self loop 0->0 plus 1->0; node 1 has no in-edge
KAT3 = dict(
    x=[[2., 4.], [6., 8.]],
    edge_index=[[0, 1], [0, 0]],
    w_l=[[1., 1.]],
    b_l=[0.5],
    w_r=[[1., -1.]],
    agg=[[4., 6.], [0., 0.]],
    out=[[8.5], [-1.5]],
)

# ---- KAT-4: F.normalize rows (eps 1e-12); the zero row stays zero
KAT4 = dict(
    x=[[3., 4.], [0., 0.], [0., -2.]],
    y=[[0.6, 0.8], [0., 0.], [0., -1.]],
)

# ---- KAT-5: skip-gram loss by hand: e0=[1,0], e1=[0,1]; t=0, c=0, n=[1], s=1
#   pos = 1, neg = 0 -> L = softplus(-1) + ln 2
KAT5 = dict(
    emb=[[1., 0.], [0., 1.]],
    targets=[0], contexts=[0], negatives=[[1]], scale=1.0,
    loss=math.log1p(math.exp(-1.0)) + math.log(2.0),
    pos_prob=1.0 / (1.0 + math.exp(-1.0)), neg_prob=0.5,
)
