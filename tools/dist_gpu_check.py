"""Two (or more) ranks on ONE GPU over gloo: every rank trains on its shard of the same global
batch and must end with the parameters a single process gets from the whole batch.  Launched by
tests/test_gpu_dist.py through torch.distributed.run; rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from graphsage_amd import GraphSAGE, TrainStep, shard_bounds  # noqa: E402
from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph  # noqa: E402


DROPOUT = float(os.environ.get("GSAGE_DIST_CHECK_DROPOUT", "0.3"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    n, B, K, steps = 20000, 4099, 5, 3      # 4099: uneven shards; n large enough for the
    # touched-row hand-over to be taken (ops.live_rows_worthwhile) on shards and whole batch alike
    ei = powerlaw_graph(n, 12, seed=3, device=dev)
    x = memory_features(n, seed=3, device=dev)
    batches = [cooccurrence_batch(ei, n, B, K, seed=10 + s) for s in range(steps)]
    torch.manual_seed(0)
    # the reference's dropout (graphsage.py:135): the in-kernel masks of step s must be the same on
    # every replica, and the same as in the single-process run below
    model = GraphSAGE(7, 64, 64, 2, DROPOUT).to(dev)
    init = {k: v.clone() for k, v in model.state_dict().items()}
    if rank == 1:
        torch.rand(1000)                 # one rank's CPU generator runs ahead: must not matter
    ts = TrainStep(model, x, ei)         # base seed: rank 0's draw, broadcast here
    assert ts.world == world
    if rank != 0:
        torch.rand(77 * rank)            # ... nor afterwards
    lo, hi = shard_bounds(B, rank, world)
    losses = []
    for t, c, neg in batches:
        stats = ts.step(t[lo:hi], c[lo:hi], neg[lo:hi], global_batch=B)
        losses.append(float(stats[0]))
    if rank == 0:
        # single-process run of the same steps on the whole batches (no process group in use)
        ref_model = GraphSAGE(7, 64, 64, 2, DROPOUT).to(dev)
        ref_model.load_state_dict(init)
        ref = TrainStep(ref_model, x, ei, seed=ts.base_seed, distributed=False)
        ref_losses = [float(ref.step(*b)[0]) for b in batches]
        worst = max(float((a.detach() - b.detach()).abs().max())
                    for a, b in zip(model.parameters(), ref_model.parameters()))
        from graphsage_amd import ops
        result = {"world": world, "losses": losses, "ref_losses": ref_losses,
                  "max_param_diff": worst, "live_rows_used": ops.LIVE_STATS["used"],
                  "dropout": DROPOUT}
    # ---- the same with the step captured: per rank two HIP graphs around the one all-reduce
    Bc = 4096                                             # even shards: the graphs fix the batch size
    cbatches = [cooccurrence_batch(ei, n, Bc, K, seed=30 + s) for s in range(steps)]
    cmodel = GraphSAGE(7, 64, 64, 2, DROPOUT).to(dev)
    cmodel.load_state_dict(init)
    cts = TrainStep(cmodel, x, ei, seed=4242)
    lo, hi = shard_bounds(Bc, rank, world)
    cts.capture_step(hi - lo, K, preserve_state=True)
    closses = [float(cts.step_captured(t[lo:hi], c[lo:hi], neg[lo:hi])[0]) for t, c, neg in cbatches]
    if rank == 0:
        rmodel = GraphSAGE(7, 64, 64, 2, DROPOUT).to(dev)
        rmodel.load_state_dict(init)
        rts = TrainStep(rmodel, x, ei, seed=4242, distributed=False)
        rts.capture_step(Bc, K, preserve_state=True)
        rlosses = [float(rts.step_captured(*b)[0]) for b in cbatches]
        result.update({"captured_losses": closses, "captured_ref_losses": rlosses,
                       "captured_max_param_diff": max(float((a.detach() - b.detach()).abs().max())
                                                      for a, b in zip(cmodel.parameters(), rmodel.parameters()))})
        print(json.dumps(result), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
