"""Generates the golden fixtures under tests/golden/ by RUNNING the reference's own Python
in this container (it never travels to the GPU box; only these small data files do).

  loss_golden.npz  -- inputs/outputs of the reference `SkipGramLoss` (src/model/loss.py),
                      loaded by file path (its package __init__ needs torch_geometric).
  alias_golden.npz -- `DegreeBiasedNegativeSampler` tables (src/walks/negative_sampler.py) on
                      the two graphs of the reference tests (tests/test_walks.py:147-155,173-177),
                      and the pair list of tests/test_walks.py:98-118.
  c1_graph.npz     -- the default 500-memory synthetic graph (seed 42): undirected edge_index
                      from GraphSAGEViewExtractor and the 7-dim features of
                      MemoryFeatureExtractor (src/data/*, loaded through a stub package
                      because src/data/__init__.py imports torch_geometric via dataset.py).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def make_loss_golden():
    ref = load_by_path("ref_loss", f"{REF}/src/model/loss.py")
    cases = [  # (name, N, D, B, K, scale, seed, special)
        ("t32", 100, 64, 32, 5, 5.0, 0, None),       # tests/test_model.py:110-124
        ("t16", 50, 64, 16, 5, 5.0, 1, None),        # tests/test_model.py:163-172
        ("b1024", 500, 64, 1024, 5, 5.0, 2, None),   # C1 batch
        ("d128k3", 500, 128, 1000, 3, 5.0, 3, None),
        ("tiny", 7, 64, 5, 1, 5.0, 4, None),
        ("dup", 60, 64, 48, 5, 5.0, 5, "dup"),       # all targets equal
        ("scale1", 100, 64, 32, 5, 1.0, 6, None),
        ("d32", 80, 32, 40, 5, 5.0, 7, None),
        ("d30k7", 80, 30, 33, 7, 5.0, 8, None),      # D % 4 != 0, K > 5
        ("unnorm", 100, 64, 64, 5, 5.0, 9, "unnorm"),  # large scores: saturating sigmoid
    ]
    out = {}
    for name, N, D, B, K, scale, seed, special in cases:
        torch.manual_seed(seed)
        emb = torch.randn(N, D)
        if special != "unnorm":
            emb = torch.nn.functional.normalize(emb, p=2, dim=1)
        else:
            emb = emb * 2.0
        emb = emb.detach().clone().requires_grad_(True)
        t = torch.randint(0, N, (B,))
        if special == "dup":
            t = torch.full((B,), 3, dtype=torch.int64)
        c = torch.randint(0, N, (B,))
        n = torch.randint(0, N, (B, K))
        fn = ref.SkipGramLoss(scale_factor=scale)
        loss, det = fn.forward_with_details(emb, t, c, n)
        loss2 = fn.forward(emb, t, c, n)
        assert torch.equal(loss, loss2)
        loss.backward()
        out[f"{name}.emb"] = emb.detach().numpy()
        out[f"{name}.targets"] = t.numpy()
        out[f"{name}.contexts"] = c.numpy()
        out[f"{name}.negatives"] = n.numpy()
        out[f"{name}.scale"] = np.float32(scale)
        out[f"{name}.loss"] = np.float32(loss.item())
        out[f"{name}.details"] = np.array(
            [det[k] for k in ("total_loss", "pos_loss", "neg_loss", "pos_score_mean",
                              "neg_score_mean", "pos_prob", "neg_prob")], dtype=np.float64)
        out[f"{name}.grad"] = emb.grad.numpy()
    out["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, "loss_golden.npz"), **out)
    print("loss_golden.npz:", len(cases), "cases")


def make_alias_golden():
    sys.path.insert(0, REF)
    from src.walks import DegreeBiasedNegativeSampler, CooccurrencePairSampler
    out = {}
    # tests/test_walks.py:147-155
    g1 = torch.tensor([[0, 0, 0, 1, 2, 3], [1, 2, 3, 0, 0, 0]], dtype=torch.int64)
    # tests/test_walks.py:173-177 (node 0 has many edges)
    g2 = torch.tensor([[0, 0, 0, 0, 0, 1, 2], [1, 2, 3, 4, 5, 0, 0]], dtype=torch.int64)
    torch.manual_seed(11)
    g3 = torch.randint(0, 40, (2, 300))
    for name, ei, n in (("g1", g1, 4), ("g2", g2, 6), ("g3", g3, 40)):
        s = DegreeBiasedNegativeSampler(ei, n, exponent=0.75)
        out[f"{name}.edge_index"] = ei.numpy()
        out[f"{name}.num_nodes"] = np.int64(n)
        out[f"{name}.probs"] = s.probs.cpu().numpy()
        out[f"{name}.prob_table"] = s.prob_table.cpu().numpy()
        out[f"{name}.alias"] = s.alias.cpu().numpy()
    ps = CooccurrencePairSampler(context_window=2)
    pairs = ps.extract_pairs([[0, 1, 2, 3, 4]])
    out["pairs_w2"] = np.array(pairs, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "alias_golden.npz"), **out)
    print("alias_golden.npz: 3 graphs,", len(pairs), "pairs")


def make_c1_graph():
    pkg = types.ModuleType("refdata")
    pkg.__path__ = [f"{REF}/src/data"]
    sys.modules["refdata"] = pkg
    gl = importlib.import_module("refdata.graph_loader")
    ve = importlib.import_module("refdata.view_extractor")
    fe = importlib.import_module("refdata.feature_extractor")
    full = gl.GraphLoader.create_mock(num_memories=500, num_entities=100, seed=42)
    view = ve.GraphSAGEViewExtractor(full)
    v = view.extract_with_undirected_edges()
    edge_index = v.edge_index
    feats = fe.MemoryFeatureExtractor(full, v.node_mapping).extract(edge_index)
    np.savez_compressed(os.path.join(OUT, "c1_graph.npz"),
                        edge_index=edge_index.numpy().astype(np.int64),
                        features=feats.numpy().astype(np.float32))
    print("c1_graph.npz: N =", feats.shape[0], "E =", edge_index.shape[1])


if __name__ == "__main__":
    make_loss_golden()
    make_alias_golden()
    make_c1_graph()
