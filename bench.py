#!/usr/bin/env python
"""bench.py -- trained co-occurrence pairs/sec of the GraphSAGE hot path on MI355X.

A step = the reference trainer's loop body (src/training/trainer.py:213-240): zero_grad,
full-graph forward (2x SAGEConv, ReLU+dropout, L2 norm), skip-gram loss on this GPU's shard of
pairs, backward, [gradient all-reduce over RCCL], clip_grad_norm_(1.0), Adam.  Graph, features
and weights are replicated per GPU; pairs are sharded (weak scaling: pairs per GPU fixed).

  python bench.py [--gpus N --steps K --warmup W] [--workload c4|c3|c2] [--pairs-per-gpu B]

Prints ONE JSON line on rank 0 (contract in the task statement), including `roofline` for the
dominant kernel (the F=64 segment gather of SAGEConv layer 2, forward + transposed backward)
and `cpu_baseline` (the CPU oracle timed on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (nodes, avg directed degree, hidden, out, description)
    "c2": (100_000, 20, 64, 64, "C2: 100k-node/~2M-edge power-law graph, SAGE 7->64->64"),
    "c3": (1_000_000, 20, 128, 128, "C3: 1M-node/~20M-edge power-law graph, SAGE 7->128->128"),
    "c4": (10_000_000, 20, 64, 64, "C4: 10M-node/~200M-edge power-law graph, SAGE 7->64->64"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (spec)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def measured_traffic(workload, F):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_hbm.json: separate FETCH_SIZE and WRITE_SIZE runs of this same command).
    gfx950 correction: FETCH_SIZE counts 128-B read requests as 64 B, so it is doubled."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc_hbm.json"))):
        best = path
    if best is None:
        return None, None
    try:
        z = json.load(open(best))
        k = f"seg_gather_vec_k<{F // 4}, 0, false>"      # the plain (unmasked) gather
        fetch_kb, write_kb = z["FETCH_SIZE"][k]["mean_KB"], z["WRITE_SIZE"][k]["mean_KB"]
        return (2.0 * fetch_kb + write_kb) * 1024.0, os.path.relpath(best, ROOT)
    except (KeyError, ValueError, OSError):
        return None, None


def cpu_baseline(args, hidden, out_dim, K):
    """The CPU oracle (oracle/sage_oracle.py: the op sequence PyG + loss.py execute on CPU,
    plus clip + Adam) on a bounded sample: the same generator at `--cpu-nodes` nodes with the
    pair batch scaled by the same factor, so pairs per node per step match the GPU workload."""
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    from oracle import sage_oracle as O
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    n = args.cpu_nodes
    nodes_full = WORKLOADS[args.workload][0]
    b = max(1024, int(args.pairs_per_gpu * n / nodes_full))
    ei = powerlaw_graph(n, WORKLOADS[args.workload][1], seed=42)
    x = memory_features(n, seed=42)
    torch.manual_seed(0)
    model = O.OracleGraphSAGE(7, hidden, out_dim, 2, 0.3).train()
    opt = O.make_adam(model)
    batches = [cooccurrence_batch(ei, n, b, K, seed=s) for s in range(2)]
    O.train_step(model, opt, x, ei, *batches[0])                 # warm-up
    steps, t0 = 0, time.perf_counter()
    while steps < args.cpu_steps:
        O.train_step(model, opt, x, ei, *batches[steps % 2])
        steps += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = (time.perf_counter() - t0) / steps
    return {
        "value": b / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
        "sample": f"{n}-node/{ei.size(1)}-edge graph from the same generator (1/{nodes_full // n} "
                  f"of the workload), {b} pairs/step (same pairs per node), {steps} steps, "
                  f"{dt * 1e3:.1f} ms/step",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c4")
    ap.add_argument("--pairs-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--negatives", type=int, default=5)
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--cpu-nodes", type=int, default=200_000)
    ap.add_argument("--cpu-steps", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--receptive-field", action="store_true",
                    help="opt-in extra (not the headline): compute only the 2-hop closure of each "
                         "batch (pays for small batches on large graphs)")
    ap.add_argument("--cache-input-aggregation", action="store_true",
                    help="opt-in extra (not the headline): keep mean_j x_j of the constant input "
                         "features across steps instead of recomputing it like the reference")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal switch for a one-GPU box: all ranks on cuda:0 over gloo (RCCL needs one GPU per rank)
    rehearse = os.environ.get("GSAGE_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        import faulthandler
        faulthandler.dump_traceback_later(80, exit=True)
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
    if args.gpus != world:
        log(f"note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from graphsage_amd import GraphSAGE, TrainStep, default_cache, ops
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph

    nodes, deg, hidden, out_dim, desc = WORKLOADS[args.workload]
    K, B = args.negatives, args.pairs_per_gpu

    t0 = time.perf_counter()
    ei = powerlaw_graph(nodes, deg, seed=42, device=dev)       # same seed -> identical replicas
    x = memory_features(nodes, seed=42, device=dev)
    E = ei.size(1)
    torch.cuda.synchronize()
    log(f"graph: N={nodes} E={E} generated in {time.perf_counter() - t0:.1f}s")
    t0 = time.perf_counter()
    graph = default_cache().get(ei, nodes)
    torch.cuda.synchronize()
    log(f"CSR build (both orientations): {time.perf_counter() - t0:.3f}s, "
        f"hub rows {graph.n_long.tolist()}, max in-degree {int(graph.in_degree().max())}")

    torch.manual_seed(0)
    model = GraphSAGE(7, hidden, out_dim, 2, args.dropout,
                      cache_input_aggregation=args.cache_input_aggregation).to(dev)
    step = TrainStep(model, x, ei, receptive_field=args.receptive_field)
    nbatch = 4
    batches = [cooccurrence_batch(ei, nodes, B, K, seed=1000 * (rank + 1) + i) for i in range(nbatch)]
    torch.cuda.synchronize()
    if rehearse:
        print(f"[bench r{rank}] batches ready", file=sys.stderr, flush=True)

    coll_dev = torch.device("cpu") if rehearse else dev      # gloo rehearsal: host tensors
    token = torch.zeros(1, device=coll_dev)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:                      # barrier = a one-element all-reduce
            dist.all_reduce(token)
            torch.cuda.synchronize()
        if rehearse:
            print(f"[bench r{rank}] barrier passed", file=sys.stderr, flush=True)

    for i in range(args.warmup):
        step.step(*batches[i % nbatch], global_batch=B * world)
        if rehearse:
            torch.cuda.synchronize()
            print(f"[bench r{rank}] warmup step {i} done", file=sys.stderr, flush=True)
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        stats = step.step(*batches[i % nbatch], global_batch=B * world)
        if rehearse:
            print(f"[bench r{rank}] timed step {i} queued", file=sys.stderr, flush=True)
    sync_all()
    elapsed = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    loss = float(stats[0].item())

    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    # roofline of the dominant kernel: the plain F=hidden segment gather (layer-2 forward, and its
    # transposed backward when that runs without a zero-row mask).  The masked backward gather
    # fetches only the rows the batch touched, i.e. fewer than the algorithmic bytes: it is timed
    # under its own tag and reported beside the roofline, never averaged into it.
    F = hidden
    launches, mean_ms = timer.summary().get(f"segment_gather_F{F}", (0, float("nan")))
    skip_launches, skip_ms = timer.summary().get(f"segment_gather_F{F}_skip", (0, float("nan")))
    alg_bytes = 4 * (nodes + 1) + 4 * E + 4 * E * F + 4 * nodes * F
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9 if launches else float("nan")
    traffic, traffic_src = measured_traffic(args.workload, F)
    roofline = {"bound": "hbm", "kernel": f"seg_gather_vec_k<{F // 4},0,false> (hub rows included), F={F}",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "launches_timed": launches, "mean_ms": mean_ms, "algorithmic_bytes": alg_bytes}
    if skip_launches:
        roofline["masked_backward_gather"] = {
            "kernel": f"seg_gather_vec_k<{F // 4},0,masked>: rows whose gradient is zero are not fetched",
            "launches_timed": skip_launches, "mean_ms": skip_ms}

    # fp32 storage and accumulation everywhere; the 128/256-wide GEMMs (C3) feed the bf16 matrix
    # pipe with each fp32 operand split exactly into three bf16 parts (fp32-equivalent results)
    dtype = "f32" if hidden < 128 or os.environ.get("GSAGE_WIDE_GEMM", "")[:1] == "f" else \
        "f32 (wide GEMMs: bf16 MFMA on 3-way split fp32 operands, fp32 accumulate)"
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle on the bounded sample ...")
            cpu = cpu_baseline(args, hidden, out_dim, K)
        print(json.dumps({
            "metric": "co-occurrence pairs/sec trained (fwd+bwd+step)", "value": value,
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc, "nodes": nodes, "edges": E, "negatives": K,
                       "pairs_per_gpu": B, "global_batch": B * world, "dropout": args.dropout,
                       "graph_pass": "2-hop closure of the batch only (opt-in)" if args.receptive_field
                       else "full graph every step (as the reference)",
                       "input_aggregation": "cached across steps (opt-in, not the reference's work)"
                       if args.cache_input_aggregation else "recomputed every step",
                       "pairs": "target = source of a random edge, context = its 1-hop neighbour or "
                                "(p = 1/2) a 2-hop one, negatives ~ out_degree^0.75",
                       "parallelism": f"replicated graph, pair-sharded dp{world}",
                       "loss_backward": "deterministic (sorted)" if ops.loss_bwd_deterministic(nodes, B, K)
                       else "atomic",
                       "final_loss": loss},
            "roofline": roofline, "cpu_baseline": cpu,
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
