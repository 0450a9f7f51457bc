"""CPU: feeder rows: negative sampler and batch generator host logic, pinned by
golden tables produced with the reference's own src/walks code (tests/golden/alias_golden.npz)."""
import os

import numpy as np
import torch

from oracle import sage_oracle as O


def test_alias_tables_bit_identical_to_reference(golden_dir):
    from graphsage_amd import DegreeBiasedNegativeSampler
    z = np.load(os.path.join(golden_dir, "alias_golden.npz"))
    for g in ("g1", "g2", "g3"):
        s = DegreeBiasedNegativeSampler(torch.from_numpy(z[f"{g}.edge_index"]), int(z[f"{g}.num_nodes"]),
                                        exponent=0.75)
        assert np.array_equal(s.probs.numpy(), z[f"{g}.probs"]), g
        assert np.array_equal(s.prob_table.numpy(), z[f"{g}.prob_table"]), g
        assert np.array_equal(s.alias.numpy(), z[f"{g}.alias"]), g
        assert s.alias.dtype == torch.int64 and s.prob_table.dtype == torch.float32


def test_alias_table_matches_oracle_on_powerlaw_graph():
    from graphsage_amd import DegreeBiasedNegativeSampler
    from graphsage_amd.synth import powerlaw_graph
    n = 50_000
    ei = powerlaw_graph(n, 12, seed=3)
    s = DegreeBiasedNegativeSampler(ei, n)
    probs, prob_table, alias = O.alias_table(ei, n)
    assert torch.equal(s.probs, probs) and torch.equal(s.prob_table, prob_table) and torch.equal(s.alias, alias)
    # the table reproduces the target distribution exactly (alias-method identity)
    recon = s.prob_table.double().clone()
    recon.index_add_(0, s.alias, (1.0 - s.prob_table.double()))
    assert torch.allclose(recon / n, s.probs.double(), rtol=1e-3, atol=1e-9)   # float32 table updates


def test_sampler_shape_range_and_degree_bias():
    """tests/test_walks.py:157-186 of the reference, on the drop-in class."""
    from graphsage_amd import DegreeBiasedNegativeSampler, UniformNegativeSampler
    ei = torch.tensor([[0, 0, 0, 1, 2, 3], [1, 2, 3, 0, 0, 0]])
    s = DegreeBiasedNegativeSampler(ei, 4, exponent=0.75)
    neg = s.sample(num_samples=10, num_negatives=5)
    assert neg.shape == (10, 5) and neg.dtype == torch.int64 and neg.min() >= 0 and neg.max() < 4
    ei2 = torch.tensor([[0, 0, 0, 0, 0, 1, 2], [1, 2, 3, 4, 5, 0, 0]])
    s2 = DegreeBiasedNegativeSampler(ei2, 6)
    torch.manual_seed(0)
    draws = s2.sample(10000, 1).flatten()
    counts = torch.bincount(draws, minlength=6).float() / 10000
    assert counts[0] > 1.0 / 6                       # node 0 drawn more often than uniform
    assert torch.allclose(counts, s2.probs, atol=0.02)
    assert abs(s2.get_probability(0) - s2.probs[0].item()) < 1e-12
    u = UniformNegativeSampler(7).sample(4, 3)
    assert u.shape == (4, 3) and u.max() < 7
    empty = DegreeBiasedNegativeSampler(torch.zeros(2, 0, dtype=torch.int64), 5)
    assert torch.allclose(empty.probs, torch.full((5,), 0.2))


def test_batch_generator_epoch_semantics():
    """batch_generator.py:140-194: every pair once per epoch, last batch smaller, drop_last."""
    from graphsage_amd import BatchGenerator, UniformNegativeSampler
    pairs = [(i, 1000 + i) for i in range(103)]                   # the reference's list layout
    gen = BatchGenerator(pairs, UniformNegativeSampler(50), batch_size=32, num_negatives=5, device="cpu")
    assert len(gen) == 4
    batches = list(gen)
    assert [b.size for b in batches] == [32, 32, 32, 7]
    t = torch.cat([b.targets for b in batches])
    c = torch.cat([b.contexts for b in batches])
    assert sorted(t.tolist()) == list(range(103)) and torch.equal(c, t + 1000)   # pairs stay paired
    assert all(b.negatives.shape == (b.size, 5) and b.negatives.dtype == torch.int64 for b in batches)
    t2 = torch.cat([b.targets for b in gen])
    assert not torch.equal(t, t2)                                # reshuffled every epoch
    gen_d = BatchGenerator(pairs, UniformNegativeSampler(50), 32, 5, device="cpu", drop_last=True)
    assert len(gen_d) == 3 and [b.size for b in gen_d] == [32, 32, 32]
    # tensor layouts of pairs.pt (generate_data.py:190-193) are accepted too
    tt = torch.arange(103)
    for form in ((tt, tt + 1000), torch.stack([tt, tt + 1000]), torch.stack([tt, tt + 1000], 1),
                 {"targets": tt, "contexts": tt + 1000}):
        g2 = BatchGenerator(form, UniformNegativeSampler(50), 64, 2, device="cpu", shuffle=False)
        b0 = next(iter(g2))
        assert torch.equal(b0.targets, tt[:64]) and torch.equal(b0.contexts, tt[:64] + 1000)


def test_split_pairs_train_val():
    """batch_generator.py:312-336: split at int(P * (1 - val_fraction)), disjoint, seeded."""
    from graphsage_amd import split_pairs_train_val
    t, c = torch.arange(1000), torch.arange(1000) + 5000
    (tt, tc), (vt, vc) = split_pairs_train_val(t, c, 0.1, seed=42)
    assert tt.numel() == 900 and vt.numel() == 100
    assert torch.equal(tc, tt + 5000) and torch.equal(vc, vt + 5000)
    assert sorted(torch.cat([tt, vt]).tolist()) == list(range(1000))
    (tt2, _), _ = split_pairs_train_val(t, c, 0.1, seed=42)
    (tt3, _), _ = split_pairs_train_val(t, c, 0.1, seed=43)
    assert torch.equal(tt, tt2) and not torch.equal(tt, tt3)


def test_pair_window_golden_shape(golden_dir):
    """The reference's window rule (tests/test_walks.py:98-118): walk [0..4], window 2 -> 14
    ordered pairs; kept as a fixture for the pair-extraction row (8f-4)."""
    z = np.load(os.path.join(golden_dir, "alias_golden.npz"))
    pairs = {tuple(p) for p in z["pairs_w2"].tolist()}
    assert len(z["pairs_w2"]) == 14
    assert (0, 1) in pairs and (0, 2) in pairs and (0, 3) not in pairs and (2, 0) in pairs


def test_oracle_pair_window_matches_reference_golden(golden_dir):
    """oracle.extract_pairs reproduces the pair LIST (order included) the reference's own
    CooccurrencePairSampler produced for tests/test_walks.py:98-118."""
    from oracle import sage_oracle as O
    z = np.load(os.path.join(golden_dir, "alias_golden.npz"))
    got = O.extract_pairs([[0, 1, 2, 3, 4]], 2)
    assert [list(p) for p in got] == z["pairs_w2"].tolist()
    # pair_sampler.py:287-305: every position sees at most 2*window others
    assert len(O.extract_pairs([[3]], 5)) == 0
    assert len(O.extract_pairs([list(range(30))], 10)) <= 30 * 20


def _stream_walks(z):
    out, p = [], 0
    flat = z["walk_flat"].tolist()
    for n in z["walk_len"].tolist():
        out.append(flat[p:p + n])
        p += n
    return out


def test_oracle_pair_stream_matches_reference_streaming_generator(golden_dir):
    """stream_golden.npz = the batches the reference's own StreamingBatchGenerator
    (batch_generator.py:205-309) produced: its stream is the pair list of the walks taken in the
    shuffled order, cut every batch_size pairs.  Pins oracle.extract_pairs on ragged walks."""
    z = np.load(os.path.join(golden_dir, "stream_golden.npz"))
    walks = _stream_walks(z)
    for tag in ("a", "b", "c"):
        w, bs = int(z[f"{tag}.window"]), int(z[f"{tag}.batch_size"])
        pairs = O.extract_pairs([walks[i] for i in z[f"{tag}.order"].tolist()], w)
        assert [p[0] for p in pairs] == z[f"{tag}.targets"].tolist()
        assert [p[1] for p in pairs] == z[f"{tag}.contexts"].tolist()
        sizes = z[f"{tag}.sizes"].tolist()
        assert sizes == [bs] * (len(pairs) // bs) + ([len(pairs) % bs] if len(pairs) % bs else [])
        est = sum(len(x) * min(2 * w, len(x) - 1) for x in walks)
        assert est == int(z[f"{tag}.estimated_pairs"])


def test_training_data_directory_round_trip(tmp_path):
    """features.pt / edge_index.pt / pairs.pt / metadata.json / walks/walks.json as
    scripts/generate_data.py:175-212 writes them and scripts/train.py:74-99 reads them."""
    import json
    from graphsage_amd import load_training_data, load_walks, save_training_data
    x = torch.rand(30, 7)
    ei = torch.randint(0, 30, (2, 90))
    t, c = torch.randint(0, 30, (200,)), torch.randint(0, 30, (200,))
    d = str(tmp_path / "processed")
    save_training_data(d, x, ei, t, c, metadata={"seed": 42}, walks=[[0, 1, 2], [3, 4]])
    # the reference's own reader lines (train.py:79-88), weights_only for safety
    assert torch.equal(torch.load(os.path.join(d, "features.pt"), weights_only=True), x)
    assert torch.equal(torch.load(os.path.join(d, "edge_index.pt"), weights_only=True), ei)
    pd = torch.load(os.path.join(d, "pairs.pt"), weights_only=True)
    assert set(pd) == {"targets", "contexts"} and pd["targets"].dtype == torch.int64
    meta = json.load(open(os.path.join(d, "metadata.json")))
    assert meta["num_nodes"] == 30 and meta["num_edges"] == 90 and meta["seed"] == 42
    fx, fe, (ft, fc), fm = load_training_data(d, device="cpu")
    assert torch.equal(fx, x) and torch.equal(fe, ei) and torch.equal(ft, t) and torch.equal(fc, c)
    assert fm == meta and load_walks(d) == [[0, 1, 2], [3, 4]]
    # and a directory written the reference's way (pairs as int32 lists turned tensors) loads too
    torch.save({"targets": t.to(torch.int32), "contexts": c.to(torch.int32)}, os.path.join(d, "pairs.pt"))
    assert load_training_data(d, device="cpu")[2][0].dtype == torch.int64
