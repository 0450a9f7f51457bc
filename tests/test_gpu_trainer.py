"""GPU: training harness: epoch means vs the oracle on injected batches, the
captured-HIP-graph step vs the eager step, checkpoints interchangeable with plain torch Adam."""
import pytest
import torch

from oracle import sage_oracle as O
from util import assert_close

pytestmark = pytest.mark.gpu


class FixedNegatives:
    """Sampler stub handing out pre-drawn negatives in order (so both sides see the same)."""

    def __init__(self, negs):
        self.negs, self.i = negs, 0

    def sample(self, b, k):
        out = self.negs[self.i][:b, :k]
        self.i += 1
        return out

    def to(self, device):
        self.negs = [n.to(device) for n in self.negs]
        return self


def _setup(n=1500, pairs=2500, hidden=64, dropout=0.0, seed=0):
    from graphsage_amd import GraphSAGE
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    ei = powerlaw_graph(n, 10, seed=42)
    x = memory_features(n, seed=42)
    t, c, _ = cooccurrence_batch(ei, n, pairs, 5, seed=7)
    torch.manual_seed(seed)
    ref = O.OracleGraphSAGE(7, hidden, hidden, 2, dropout)
    m = GraphSAGE(7, hidden, hidden, 2, dropout)
    m.load_state_dict(ref.state_dict())
    return n, ei, x, t, c, m.cuda(), ref


def test_epoch_means_match_oracle_on_injected_batches():
    from graphsage_amd import GraphSAGETrainer
    n, ei, x, t, c, m, ref = _setup()
    bs = 1024
    g = torch.Generator().manual_seed(1)
    negs = [torch.randint(0, n, (2 * bs, 5), generator=g) for _ in range(8)]
    cfg = {"training": {"batch_size": bs, "learning_rate": 1e-3, "gradient_clip": 1.0},
           "negatives": {"per_positive": 5}, "validation": {"val_fraction": 0.0}}
    tr = GraphSAGETrainer(m, x.cuda(), ei.cuda(), (t, c), FixedNegatives(list(negs)), cfg)
    tr.batch_generator.shuffle = False
    assert len(tr.batch_generator) == 3                       # 1024, 1024, 452 (drop_last=False)
    metrics = tr.train_epoch()
    opt = O.make_adam(ref)
    ref.train()
    acc = {"loss": 0.0, "pos_loss": 0.0, "neg_loss": 0.0}
    for i, lo in enumerate(range(0, t.numel(), bs)):
        hi = min(lo + bs, t.numel())
        _, d = O.train_step(ref, opt, x, ei, t[lo:hi], c[lo:hi], negs[i][:hi - lo])
        acc["loss"] += d["total_loss"]; acc["pos_loss"] += d["pos_loss"]; acc["neg_loss"] += d["neg_loss"]
    for k in acc:
        assert metrics[k] == pytest.approx(acc[k] / 3, rel=2e-5), k
    for (name, pg), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        assert_close(pg, pr, rtol=1e-4, floor=2e-5, what=name)


def test_trainer_with_and_without_captured_step_agree():
    """Full batches replayed from one captured HIP graph (the default on small graphs) against
    the eager step: same epoch means and parameters (dropout 0, so no random stream differs)."""
    from graphsage_amd import GraphSAGETrainer
    out = []
    for hg in (True, False):
        n, ei, x, t, c, m, _ = _setup()
        g = torch.Generator().manual_seed(1)
        negs = [torch.randint(0, n, (2048, 5), generator=g) for _ in range(12)]
        cfg = {"training": {"batch_size": 1024, "hip_graph": hg}, "negatives": {"per_positive": 5},
               "validation": {"val_fraction": 0.0}}
        tr = GraphSAGETrainer(m, x.cuda(), ei.cuda(), (t, c), FixedNegatives(list(negs)), cfg)
        tr.batch_generator.shuffle = False
        assert tr._use_graph is hg
        hist = [tr.train_epoch() for _ in range(3)]
        out.append((hist, [p.detach().clone() for p in m.parameters()]))
    for a, b in zip(out[0][0], out[1][0]):
        for k in a:
            assert a[k] == pytest.approx(b[k], rel=1e-5), k
    for pa, pb in zip(out[0][1], out[1][1]):
        assert_close(pa, pb, rtol=1e-4, floor=2e-5)


def test_validate_and_train_loop_and_checkpoint(tmp_path):
    from graphsage_amd import DegreeBiasedNegativeSampler, GraphSAGE, GraphSAGETrainer
    n, ei, x, t, c, m, ref = _setup(dropout=0.3)
    cfg = {"training": {"batch_size": 512, "early_stopping_patience": 50},
           "negatives": {"per_positive": 5}, "validation": {"val_fraction": 0.1},
           "paths": {"checkpoints": str(tmp_path / "ck")}}
    sampler = DegreeBiasedNegativeSampler(ei, n)
    tr = GraphSAGETrainer(m, x.cuda(), ei.cuda(), list(zip(t.tolist(), c.tolist())), sampler, cfg)
    assert tr.train_pairs[0].numel() == 2250 and tr.val_pairs[0].numel() == 250
    v0 = tr.validate()
    assert set(v0) == {"loss", "pos_prob", "neg_prob"} and 0 < v0["loss"] < 10
    best = tr.train(num_epochs=6)
    assert best < v0["loss"] and 0 < best < 10                   # tests/test_integration.py:93-157
    assert tr.get_embeddings().shape == (n, 64)
    ck = torch.load(tmp_path / "ck" / "model_best.pt", weights_only=True)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "best_loss", "config"} <= set(ck)
    # the reference's loaders accept it: strict model load + a plain per-parameter Adam
    ref.load_state_dict(ck["model_state_dict"], strict=True)
    opt = O.make_adam(ref)
    opt.load_state_dict(ck["optimizer_state_dict"])
    assert len(opt.state_dict()["state"]) == 6
    m2 = GraphSAGE(7, 64, 64, 2, 0.3).cuda()
    tr2 = GraphSAGETrainer(m2, x.cuda(), ei.cuda(), (t, c), sampler, cfg)
    tr2.load_checkpoint(str(tmp_path / "ck" / "model_best.pt"))
    for pa, pb in zip(m2.state_dict().values(), ck["model_state_dict"].values()):
        assert torch.equal(pa.cpu(), pb.cpu())


def test_resume_from_exported_optimizer_state_continues_identically():
    """Adam moments exported per parameter -> torch Adam on the oracle continues in step."""
    from graphsage_amd import TrainStep
    from graphsage_amd.synth import cooccurrence_batch
    n, ei, x, t, c, m, ref = _setup()
    ts = TrainStep(m, x.cuda(), ei.cuda())
    batches = [cooccurrence_batch(ei, n, 512, 5, seed=s) for s in range(4)]
    for b in batches[:2]:
        ts.step(*[v.cuda() for v in b])
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    opt = O.make_adam(ref)
    opt.load_state_dict(ts.optimizer_state_dict())
    ref.train()
    for b in batches[2:]:
        ts.step(*[v.cuda() for v in b])
        O.train_step(ref, opt, x, ei, *b)
    for (name, pg), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        assert_close(pg, pr, rtol=1e-4, floor=2e-5, what=name)


def test_captured_step_matches_eager_step():
    from graphsage_amd import GraphSAGE, TrainStep
    from graphsage_amd.synth import cooccurrence_batch
    n, ei, x, t, c, m, ref = _setup(n=500, pairs=10)
    m2 = GraphSAGE(7, 64, 64, 2, 0.0).cuda()
    m2.load_state_dict(m.state_dict())
    snap = {k: v.clone() for k, v in m.state_dict().items()}
    eager = TrainStep(m, x.cuda(), ei.cuda())
    graph = TrainStep(m2, x.cuda(), ei.cuda())
    graph.capture_step(batch_size=256, num_negatives=5, warmup=1)
    # warm-up and capture ran real steps (on all-zero pairs): rewind the captured side's
    # parameters and Adam state to the common start
    m2.load_state_dict(snap)
    st = graph.optimizer.state[graph.flat.flat_param]
    st["step"].zero_(), st["exp_avg"].zero_(), st["exp_avg_sq"].zero_()
    for s in range(4):
        b = [v.cuda() for v in cooccurrence_batch(ei, n, 256, 5, seed=20 + s)]
        se = eager.step(*b).clone()
        sg = graph.step_captured(*b).clone()
        assert_close(sg[:7], se[:7], rtol=2e-5, floor=1e-6, what=f"stats step {s}")
    for (name, pa), (_, pb) in zip(m.named_parameters(), m2.named_parameters()):
        assert_close(pb, pa, rtol=1e-4, floor=2e-5, what=name)


def test_captured_step_with_dropout_draws_new_masks_each_replay():
    from graphsage_amd import GraphSAGE, TrainStep
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n = 800
    ei = powerlaw_graph(n, 10, seed=42).cuda()
    x = memory_features(n, seed=42).cuda()
    torch.manual_seed(0)
    m = GraphSAGE(7, 64, 64, 2, 0.5).cuda()
    ts = TrainStep(m, x, ei, learning_rate=0.0)                 # lr 0: only the mask changes
    ts.capture_step(128, 5, warmup=1)
    b = [v for v in cooccurrence_batch(ei, n, 128, 5, seed=3)]
    l1 = ts.step_captured(*b)[0].item()
    l2 = ts.step_captured(*b)[0].item()
    assert l1 != l2 and abs(l1 - l2) < 1.0


@pytest.mark.parametrize("n,b,seed", [(20000, 256, 0), (3000, 64, 1), (500, 1024, 2)])
def test_receptive_field_step_equals_full_graph_step(n, b, seed):
    """TrainStep(receptive_field=True) computes only the 2-hop closure of the batch (hub rows,
    isolated nodes and duplicate ids included) and must train exactly like the full-graph step."""
    from graphsage_amd import GraphSAGE, TrainStep
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    ei = powerlaw_graph(n, 12, seed=seed, device="cuda")
    x = memory_features(n, seed=seed, device="cuda")
    torch.manual_seed(seed)
    ma = GraphSAGE(7, 64, 64, 2, 0.0).cuda()
    mb = GraphSAGE(7, 64, 64, 2, 0.0, cache_input_aggregation=(seed == 1)).cuda()   # both forms
    mb.load_state_dict(ma.state_dict())
    full, rf = TrainStep(ma, x, ei), TrainStep(mb, x, ei, receptive_field=True)
    for s in range(4):
        t, c, neg = cooccurrence_batch(ei, n, b, 5, seed=100 + s)
        if s == 1:
            neg[:, 0] = t[0]                               # a row hit by every pair
        sa = full.step(t, c, neg).clone()
        sb = rf.step(t, c, neg).clone()
        assert_close(sb[:7], sa[:7], rtol=1e-5, floor=1e-6, what=f"stats step {s}")
        assert sb[7].item() == 0
    for (name, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert_close(pb, pa, rtol=1e-4, floor=2e-5, what=name)
    with pytest.raises(IndexError):
        rf.step(t, c.clone().fill_(n), neg)


def test_captured_step_survives_cache_churn_and_refuses_modified_inputs():
    """The captured HIP graph has the addresses of the CSR / padded features baked in; they are
    owned by the process-wide GraphCache, which other graphs and feature tensors may evict
    (ADVICE r1, medium).  TrainStep holds them itself: after the cache has been churned and the
    freed memory overwritten, replays still equal a run that was left alone."""
    from graphsage_amd import TrainStep, default_cache
    from graphsage_amd.synth import memory_features, powerlaw_graph
    bs, k = 512, 5
    g = torch.Generator().manual_seed(3)
    results = []
    for churn in (False, True):
        n, ei, x, t, c, m, _ = _setup(seed=1)
        negs = torch.randint(0, n, (6, bs, k), generator=torch.Generator().manual_seed(5)).cuda()
        xg, eig = x.cuda(), ei.cuda()
        default_cache().clear()
        step = TrainStep(m, xg, eig)
        step.capture_step(bs, k, preserve_state=True)
        assert len(step._held) >= 2                         # CSR + padded features
        for i in range(6):
            if churn and i == 2:
                builds = default_cache().builds
                junk = []
                for s in range(5):                          # five other graphs and feature tensors
                    e2 = powerlaw_graph(900 + s, 6, seed=s).cuda()
                    x2 = memory_features(900 + s, seed=s).cuda()
                    default_cache().get(e2, 900 + s)
                    default_cache().padded(x2, 8)
                    junk.append((e2, x2))
                assert default_cache().builds == builds + 5
                torch.cuda.empty_cache()
                filler = [torch.full((1 << 20,), 0x7FFFFFF0, dtype=torch.int32, device="cuda")
                          for _ in range(16)]               # whatever was freed gets overwritten
                del filler
            lo = (i * bs) % (t.numel() - bs)
            step.step_captured(t[lo:lo + bs].cuda(), c[lo:lo + bs].cuda(), negs[i])
        torch.cuda.synchronize()
        results.append([p.detach().clone() for p in m.parameters()])
        if churn:
            eig.add_(0)                                     # in-place change of the graph
            with pytest.raises(RuntimeError, match="capture again"):
                step.step_captured(t[:bs].cuda(), c[:bs].cuda(), negs[0])
    for pa, pb in zip(*results):
        assert_close(pa, pb, rtol=1e-5, floor=2e-6)


def test_train_graphsage_entry_point():
    """trainer.py:408-486 ("the recommended entry point"): dataset + walks + config in, trained
    model and unit-norm embeddings out; the loss goes down and linked nodes end up closer than
    random ones."""
    from graphsage_amd import RandomWalkGenerator, train_graphsage
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 3000
    ei = powerlaw_graph(n, 8, seed=5).cuda()
    x = memory_features(n, seed=5).cuda()

    class Data:                                   # the reference's dataset protocol
        def get_data(self, device=None):
            class D:
                pass
            d = D()
            d.x, d.edge_index, d.num_nodes = x.to(device), ei.to(device), n
            return d

    walks = RandomWalkGenerator(ei, n, walk_length=10, walks_per_node=2, seed=3).generate_all_walks()
    cfg = {"features": {"dimensions": 7}, "model": {"hidden_dim": 64, "output_dim": 64, "num_layers": 2,
                                                    "dropout": 0.1},
           "walks": {"context_window": 3}, "negatives": {"per_positive": 5, "exponent": 0.75},
           "training": {"epochs": 4, "batch_size": 4096, "learning_rate": 0.01,
                        "early_stopping_patience": 10}}
    torch.manual_seed(0)
    model, emb = train_graphsage(Data(), walks, cfg, device="cuda")
    assert emb.shape == (n, 64) and torch.allclose(emb.norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)
    src, dst = ei[0, :5000], ei[1, :5000]
    linked = (emb[src] * emb[dst]).sum(1).mean()
    rnd = (emb[src] * emb[torch.randint(0, n, (5000,), device="cuda")]).sum(1).mean()
    assert float(linked - rnd) > 0.02
    # the list-of-lists form of the walks and a plain (features, edge_index) pair work too
    lists = RandomWalkGenerator.to_lists(*walks)[:500]
    cfg["training"]["epochs"] = 1
    _, emb2 = train_graphsage((x, ei), lists, cfg, device="cuda")
    assert emb2.shape == (n, 64)
