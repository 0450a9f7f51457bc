"""Writes oracle_c1_snapshot.npz: the oracle's embeddings on the C1 graph fixture with
torch.manual_seed(0) weights.  Needs no reference code."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import sage_oracle as O  # noqa: E402

z = np.load(os.path.join(HERE, "c1_graph.npz"))
torch.manual_seed(0)
m = O.OracleGraphSAGE(7, 64, 64, 2, 0.3).eval()
emb = m(torch.from_numpy(z["features"]), torch.from_numpy(z["edge_index"]))
np.savez_compressed(os.path.join(HERE, "oracle_c1_snapshot.npz"), embeddings=emb.detach().numpy())
print("oracle_c1_snapshot.npz", emb.shape)
