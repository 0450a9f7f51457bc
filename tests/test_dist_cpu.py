"""CPU, world_size 2 over gloo: the pair-sharded data-parallel step of TrainStep (shard split,
global-mean normalisation, ONE all-reduce of the flat gradient+stats buffer, replicated
clip + Adam) gives the same parameters and statistics as one process on the whole batch.
The device arithmetic is stood in for by the oracle (test infrastructure), because the HIP
kernels cannot run here; the exchange logic under test is the product's own."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import sage_oracle as O


class ShardLoss(torch.nn.Module):
    """loss.py:161-193 arithmetic with the denominators of the GLOBAL batch (what the K4
    kernel does with loss_norm_B / loss_norm_BK), returning (loss, stats[8])."""

    def __init__(self, scale=5.0):
        super().__init__()
        self.scale = scale

    def forward_stats(self, emb, t, c, n, global_batch=None):
        gb = float(global_batch if global_batch is not None else t.numel())
        k = n.size(1)
        e_t, e_c, e_n = emb[t], emb[c], emb[n]
        pos_raw = (e_t * e_c).sum(1)
        neg_raw = torch.bmm(e_n, e_t.unsqueeze(2)).squeeze(2)
        pos, neg = pos_raw * self.scale, neg_raw * self.scale
        pos_loss = -F.logsigmoid(pos).sum() / gb
        neg_loss = -F.logsigmoid(-neg).sum() / (gb * k)
        loss = pos_loss + neg_loss
        stats = torch.stack([loss, pos_loss, neg_loss, pos_raw.sum() / gb, neg_raw.sum() / (gb * k),
                             torch.sigmoid(pos).sum() / gb, torch.sigmoid(neg).sum() / (gb * k),
                             torch.zeros(())]).detach()
        return loss, stats


class StreamedDropoutOracle(O.OracleGraphSAGE):
    """The oracle model with the hook TrainStep arms before every step
    (``set_dropout_stream(base, step)``, as graphsage_amd.GraphSAGE has it): the dropout masks
    are injected (oracle ``keep_masks``) and drawn from a generator seeded by (base, step, layer)
    alone -- the stand-in for the in-kernel counter hash, which cannot run on the CPU."""

    _stream = None

    def set_dropout_stream(self, base, step=0):
        self._stream = (base, step)

    def forward(self, x, edge_index, keep_masks=None):
        from graphsage_amd.graphsage import mix_seed
        if self.training and self._stream is not None:
            keep_masks = []
            for layer in range(len(self.convs) - 1):
                g = torch.Generator().manual_seed(mix_seed(*self._stream, layer) & 0x7FFFFFFFFFFFFFFF)
                keep_masks.append((torch.rand(x.size(0), self.hidden_channels, generator=g)
                                   >= self.dropout_rate).float())
        return super().forward(x, edge_index, keep_masks)


def _inputs():
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    n = 400
    ei = powerlaw_graph(n, 8, seed=42)
    x = memory_features(n, seed=42)
    batches = [cooccurrence_batch(ei, n, 257, 5, seed=50 + s) for s in range(3)]   # odd size: uneven shards
    return n, ei, x, batches


def _run(rank, world, port, out_path, dropout=0.0):
    from graphsage_amd import TrainStep, shard_bounds
    torch.set_num_threads(1)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n, ei, x, batches = _inputs()
    torch.manual_seed(0)
    model = StreamedDropoutOracle(7, 16, 16, 2, dropout) if dropout else O.OracleGraphSAGE(7, 16, 16, 2, 0.0)
    if rank == 1:
        torch.rand(123)                  # this rank's CPU generator is somewhere else: its own
    ts = TrainStep(model, x, ei, loss_fn=ShardLoss())      # seed draw differs; rank 0's is broadcast
    assert ts.world == world and ts.rank == rank
    stats_log = []
    for t, c, neg in batches:
        if rank == 1:
            torch.rand(17)               # ... and keeps drifting between steps
        lo, hi = shard_bounds(t.numel(), rank, world)
        s = ts.step(t[lo:hi], c[lo:hi], neg[lo:hi], global_batch=t.numel())
        stats_log.append(s.clone())
    if rank == 0:
        torch.save({"params": [p.detach().clone() for p in model.parameters()], "stats": stats_log,
                    "base_seed": ts.base_seed}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(180)
def test_two_rank_step_equals_single_process():
    with tempfile.TemporaryDirectory() as d:
        one, two = os.path.join(d, "w1.pt"), os.path.join(d, "w2.pt")
        _run(0, 1, 0, one)
        mp.spawn(_run, args=(2, _free_port(), two), nprocs=2, join=True)
        a, b = torch.load(one, weights_only=True), torch.load(two, weights_only=True)
    for pa, pb in zip(a["params"], b["params"]):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
    for sa, sb in zip(a["stats"], b["stats"]):
        assert torch.allclose(sa[:7], sb[:7], rtol=1e-5, atol=1e-7)
    # and the single-process harness reproduces the reference step body (trainer.py:213-240)
    n, ei, x, batches = _inputs()
    torch.manual_seed(0)
    ref = O.OracleGraphSAGE(7, 16, 16, 2, 0.0).train()
    opt = O.make_adam(ref)
    for t, c, neg in batches:
        O.train_step(ref, opt, x, ei, t, c, neg)
    for pa, pr in zip(a["params"], ref.parameters()):
        assert torch.allclose(pa, pr.detach(), rtol=1e-5, atol=1e-7)


@pytest.mark.timeout(180)
def test_two_rank_step_with_dropout_needs_no_generator_lockstep():
    """dropout 0.3 (graphsage.py:135): the masks of step s hang on (base seed, s, layer) only; the
    base seed is rank 0's, broadcast when the stepper is built.  Rank 1 consumes extra CPU random
    numbers before and between the steps and the two-rank run still equals the single process."""
    with tempfile.TemporaryDirectory() as d:
        one, two = os.path.join(d, "w1.pt"), os.path.join(d, "w2.pt")
        _run(0, 1, 0, one, 0.3)
        mp.spawn(_run, args=(2, _free_port(), two, 0.3), nprocs=2, join=True)
        a, b = torch.load(one, weights_only=True), torch.load(two, weights_only=True)
    assert a["base_seed"] == b["base_seed"]
    for pa, pb in zip(a["params"], b["params"]):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
    for sa, sb in zip(a["stats"], b["stats"]):
        assert torch.allclose(sa[:7], sb[:7], rtol=1e-5, atol=1e-7)
    # and dropout really was on: the run differs from the dropout-free one
    with tempfile.TemporaryDirectory() as d:
        p0 = os.path.join(d, "w0.pt")
        _run(0, 1, 0, p0, 0.0)
        c = torch.load(p0, weights_only=True)
    assert not all(torch.allclose(pa, pc, rtol=1e-4, atol=1e-6) for pa, pc in zip(a["params"], c["params"]))
