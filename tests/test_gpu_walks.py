"""GPU: device random walks and co-occurrence pair windows against the oracle's
restatement of pair_sampler.py:56-86 (bit-exact, order included), the reference-generated golden
pair list, and the structural walk properties the reference tests (tests/test_walks.py:30-95)."""
import os

import numpy as np
import pytest
import torch

from oracle import sage_oracle as O

pytestmark = pytest.mark.gpu


def _graph(n=300, seed=3):
    from graphsage_amd.synth import powerlaw_graph
    return powerlaw_graph(n, 6, seed=seed)


def _adj(ei, n):
    adj = [[] for _ in range(n)]
    for s, d in ei.t().tolist():
        adj[s].append(d)
    return adj


def test_pairs_golden_through_device(golden_dir):
    from graphsage_amd import CooccurrencePairSampler
    z = np.load(os.path.join(golden_dir, "alias_golden.npz"))
    walks = torch.tensor([[0, 1, 2, 3, 4]], dtype=torch.int32, device="cuda")
    lengths = torch.tensor([5], dtype=torch.int32, device="cuda")
    t, c = CooccurrencePairSampler(2).extract_pairs_tensor(walks, lengths)
    assert t.dtype == torch.int64 and c.dtype == torch.int64
    assert torch.stack([t, c], 1).cpu().tolist() == z["pairs_w2"].tolist()


@pytest.mark.parametrize("L,window", [(1, 3), (2, 1), (7, 2), (12, 10), (40, 5), (80, 10)])
def test_pairs_match_oracle_ragged(L, window):
    """Ragged walks (lengths 1..L, single-node walks included): same pairs, same order."""
    from graphsage_amd import CooccurrencePairSampler
    g = torch.Generator().manual_seed(L * 100 + window)
    W = 257
    lengths = torch.randint(1, L + 1, (W,), generator=g, dtype=torch.int32)
    walks = torch.randint(0, 1000, (W, L), generator=g, dtype=torch.int32)
    for w in range(W):
        walks[w, lengths[w]:] = -1
    lists = [walks[w, :lengths[w]].tolist() for w in range(W)]
    want = O.extract_pairs(lists, window)
    s = CooccurrencePairSampler(window)
    t, c = s.extract_pairs_tensor(walks.cuda(), lengths.cuda())
    assert t.numel() == len(want)
    if want:
        assert torch.stack([t, c], 1).cpu().tolist() == [list(p) for p in want]
    off = s.index(walks.cuda(), lengths.cuda()).cpu().tolist()
    assert off[0] == 0 and off[-1] == len(want)
    assert [off[i + 1] - off[i] for i in range(W)] == [len(O.extract_pairs([l], window)) for l in lists]


def test_pairs_empty():
    from graphsage_amd import CooccurrencePairSampler
    walks = torch.full((4, 6), -1, dtype=torch.int32, device="cuda")
    walks[:, 0] = torch.arange(4, dtype=torch.int32)
    lengths = torch.ones(4, dtype=torch.int32, device="cuda")
    t, c = CooccurrencePairSampler(3).extract_pairs_tensor(walks, lengths)
    assert t.numel() == 0 and c.numel() == 0


def test_sampled_pairs_are_members_and_uniform():
    from graphsage_amd import CooccurrencePairSampler
    walks = torch.tensor([[0, 1, 2, 3, 4], [5, 6, -1, -1, -1], [7, -1, -1, -1, -1]],
                         dtype=torch.int32, device="cuda")
    lengths = torch.tensor([5, 2, 1], dtype=torch.int32, device="cuda")
    s = CooccurrencePairSampler(2)
    all_pairs = O.extract_pairs([[0, 1, 2, 3, 4], [5, 6]], 2)
    off = s.index(walks, lengths)
    assert int(off[-1]) == len(all_pairs) == 16
    g = torch.Generator(device="cuda").manual_seed(1)
    t, c = s.sample_pairs(walks, lengths, off, 160000, generator=g)
    got = (t * 10 + c).cpu()
    keys, counts = torch.unique(got, return_counts=True)
    assert sorted(keys.tolist()) == sorted(a * 10 + b for a, b in all_pairs)
    assert (counts.float() / 10000 - 1).abs().max() < 0.05


def test_walk_structure():
    """tests/test_walks.py:43-87: starts at the node, length <= walk_length, every transition an
    edge, isolated start -> single-node walk; plus walks_per_node walks for every node."""
    from graphsage_amd import RandomWalkGenerator
    n = 300
    ei = _graph(n)
    ei = ei[:, (ei[0] != 17) & (ei[0] != 123)]          # nodes 17 and 123 have no out-edges
    adj = _adj(ei, n)
    gen = RandomWalkGenerator(ei.cuda(), n, walk_length=20, walks_per_node=5, seed=42)
    walks, lengths = gen.generate_all_walks()
    assert walks.shape == (n * 5, 20) and walks.dtype == torch.int32
    W, Ln = walks.cpu().tolist(), lengths.cpu().tolist()
    for w, (row, k) in enumerate(zip(W, Ln)):
        assert row[0] == w // 5 and 1 <= k <= 20
        assert all(v == -1 for v in row[k:])
        for a, b in zip(row[:k - 1], row[1:k]):
            assert b in adj[a]
        if k < 20:
            assert not adj[row[k - 1]]                    # stopped only at a dead end
    assert Ln[17 * 5] == 1 and Ln[123 * 5 + 4] == 1
    lists = RandomWalkGenerator.to_lists(walks, lengths)
    assert all(len(l) > 1 for l in lists) and len(lists) == sum(k > 1 for k in Ln)
    # seeded: same seed same walks, another seed different walks
    w2, _ = RandomWalkGenerator(ei.cuda(), n, 20, 5, seed=42).generate_all_walks()
    w3, _ = RandomWalkGenerator(ei.cuda(), n, 20, 5, seed=43).generate_all_walks()
    assert torch.equal(walks, w2) and not torch.equal(walks, w3)


def test_walk_neighbour_choice_is_uniform():
    """generator.py:139 random.choice: each out-neighbour equally likely (multi-edges counted)."""
    from graphsage_amd import RandomWalkGenerator
    # node 0 -> 1,2,3,3 ; everything else -> 0
    ei = torch.tensor([[0, 0, 0, 0, 1, 2, 3], [1, 2, 3, 3, 0, 0, 0]])
    gen = RandomWalkGenerator(ei.cuda(), 4, walk_length=2, walks_per_node=40000, seed=5)
    walks, lengths = gen.generate_all_walks()
    first = walks[:40000, 1].cpu()
    frac = torch.bincount(first, minlength=4).float() / 40000
    assert abs(frac[1] - 0.25) < 0.01 and abs(frac[2] - 0.25) < 0.01 and abs(frac[3] - 0.5) < 0.01
    assert int(lengths.min()) == 2


def test_walks_to_training_pairs_end_to_end():
    """walks -> pairs -> one training step runs (the feeder rows compose with the hot path)."""
    from graphsage_amd import (CooccurrencePairSampler, DegreeBiasedNegativeSampler, GraphSAGE,
                               RandomWalkGenerator, SkipGramLoss)
    from graphsage_amd.synth import memory_features
    n = 500
    ei = _graph(n, seed=9).cuda()
    walks, lengths = RandomWalkGenerator(ei, n, 10, 2, seed=1).generate_all_walks()
    s = CooccurrencePairSampler(3)
    off = s.index(walks, lengths)
    t, c = s.sample_pairs(walks, lengths, off, 1024)
    neg = DegreeBiasedNegativeSampler(ei.cpu(), n).to("cuda").sample(1024, 5)
    m = GraphSAGE(7, 64, 64, 2, 0.0).cuda()
    loss = SkipGramLoss()(m(memory_features(n, seed=1).cuda(), ei), t, c, neg)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters())
