#!/usr/bin/env python
"""bench.py -- trained co-occurrence pairs/sec of the GraphSAGE hot path on MI355X.

A step = the reference trainer's loop body (src/training/trainer.py:213-240): zero_grad,
full-graph forward (2x SAGEConv, ReLU+dropout, L2 norm), skip-gram loss on this GPU's shard of
pairs, backward, [gradient all-reduce over RCCL], clip_grad_norm_(1.0), Adam.  Graph, features
and weights are replicated per GPU; pairs are sharded (weak scaling: pairs per GPU fixed).

  python bench.py [--gpus N --steps K --warmup W] [--workload c4|c3|c2|c5] [--pairs-per-gpu B]

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process touches no GPU;
it starts N ranks with `python -m torch.distributed.run` (one per GPU, RCCL), relays rank 0's
single JSON line and exits non-zero when any rank fails or the line does not say N ranks.

`--workload c5` is BASELINE config 5: one step = the full-graph inference encode of
src/inference/encoder.py:181-185 (eval, no_grad, 3 layers x 256 wide) under the protocol of
scripts/benchmark_latency.py:66-105; it does not shard (N > 1 = N replicas).

Prints ONE JSON line on rank 0 (contract in the task statement), including `roofline` for the
dominant kernel and `cpu_baseline` (the CPU oracle timed on a bounded sample of the workload).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (nodes, avg directed degree, hidden, out, layers, description)
    "c2": (100_000, 20, 64, 64, 2, "C2: 100k-node/~2M-edge power-law graph, SAGE 7->64->64"),
    "c3": (1_000_000, 20, 128, 128, 2, "C3: 1M-node/~20M-edge power-law graph, SAGE 7->128->128"),
    "c4": (10_000_000, 20, 64, 64, 2, "C4: 10M-node/~200M-edge power-law graph, SAGE 7->64->64"),
    "c5": (10_000_000, 20, 256, 256, 3, "C5: full-graph inference encode, 10M-node/~200M-edge "
                                        "power-law graph, SAGE 7->256->256->256 + L2 norm"),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (spec)
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 matrix peak (no sparsity)
MFMA_F32_PEAK_TFS = 157.3    # fp32-input matrix peak
SELFTEST = os.environ.get("GSAGE_BENCH_DIST_SELFTEST") == "1"   # CPU test of the launch path


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c4")
    ap.add_argument("--pairs-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--negatives", type=int, default=5)
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--nodes", type=int, default=0,
                    help="override the workload's node count (development runs; labelled)")
    ap.add_argument("--cpu-nodes", type=int, default=200_000)
    ap.add_argument("--cpu-steps", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-batch", action="store_true",
                    help="skip the short second loop at the reference's batch of 1024 pairs")
    ap.add_argument("--receptive-field", action="store_true",
                    help="opt-in extra (not the headline): compute only the 2-hop closure of each "
                         "batch (pays for small batches on large graphs)")
    ap.add_argument("--cache-input-aggregation", action="store_true",
                    help="opt-in extra (not the headline): keep mean_j x_j of the constant input "
                         "features across steps instead of recomputing it like the reference")
    ap.add_argument("--cpu-c4-one-step", action="store_true",
                    help="CPU leg only (no GPU work, minutes of host time): ONE step of the CPU oracle on "
                         "the full C4 graph at the bench's batch, SURVEY 8(d)'s last CPU figure; the "
                         "result is kept in profiles/, the default run does not repeat it")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ N > 1 without a launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """Parent of an N-rank run.  Makes NO GPU call: the ranks are fresh child processes of
    `python -m torch.distributed.run` (one per GPU, LOCAL_RANK set by the launcher).  Relays the
    one JSON line rank 0 prints; returns the exit code (non-zero if any rank failed, if there is
    not exactly one line, or if the line does not report N ranks)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:8])} ...",
          file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for raw in proc.stdout:
        raw = raw.strip()
        if raw.startswith("{") and '"metric"' in raw:
            lines.append(raw)
        elif raw:
            print(raw, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        print(f"[bench] FAILED: the {n}-rank run exited with status {rc}; no result line",
              file=sys.stderr, flush=True)
        return rc if 0 < rc < 256 else 1
    if len(lines) != 1:
        print(f"[bench] FAILED: expected one result line from rank 0, got {len(lines)}",
              file=sys.stderr, flush=True)
        return 3
    rec = json.loads(lines[0])
    if rec.get("n_gpus") != n or rec.get("ranks_seen") != n:
        print(f"[bench] FAILED: asked for {n} ranks, the line reports n_gpus={rec.get('n_gpus')} "
              f"ranks_seen={rec.get('ranks_seen')}", file=sys.stderr, flush=True)
        return 4
    print(lines[0], flush=True)
    return 0


# ------------------------------------------------------------------------- rank-side set-up
class Ranks:
    """This process's place in the job and the collectives the bench itself needs."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world_env = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # rehearsal switch for a one-GPU box: all ranks on cuda:0 over gloo (RCCL needs one GPU
        # per rank)
        self.rehearse = os.environ.get("GSAGE_BENCH_REHEARSE_ONE_GPU") == "1"
        self.backend = None
        if self.rehearse:
            import faulthandler
            faulthandler.dump_traceback_later(300, exit=True)
            self.local_rank = 0
        if self.world_env > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearse or SELFTEST:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world_env)
            else:
                if torch.cuda.device_count() <= self.local_rank:
                    raise RuntimeError(f"rank {self.rank}: LOCAL_RANK {self.local_rank} but only "
                                       f"{torch.cuda.device_count()} GPU(s) visible")
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world_env,
                                        device_id=torch.device(f"cuda:{self.local_rank}"))
            self.backend = dist.get_backend()
        self.world = dist.get_world_size() if self.world_env > 1 else 1
        if args.gpus != self.world:
            raise SystemExit(f"[bench] --gpus {args.gpus} but the process group has {self.world} "
                             f"rank(s): refusing to print a result line")
        self.dev = torch.device("cpu") if SELFTEST else torch.device(f"cuda:{self.local_rank}")
        if not SELFTEST:
            torch.cuda.set_device(self.dev)
        # gloo rehearsal / self-test: the bench's own small collectives go through host tensors
        self.coll_dev = torch.device("cpu") if (self.rehearse or SELFTEST) else self.dev
        self._token = torch.zeros(1, device=self.coll_dev)

    def device_sync(self):
        if not SELFTEST:
            self.torch.cuda.synchronize()

    def sync_all(self):
        """barrier + device synchronize on both sides (the barrier is a one-element all-reduce)."""
        self.device_sync()
        if self.world > 1:
            self.dist.all_reduce(self._token)
            self.device_sync()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def count_ranks(self):
        """Every rank adds 1: the number of processes that really took part."""
        if self.world == 1:
            return 1
        t = self.torch.ones(1, dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t)
        return int(round(float(t.item())))

    def finish(self):
        if self.world > 1:
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------ provenance
def csrc_hash():
    """sha256 over the kernel sources libgsage.so is built from (names + contents, sorted)."""
    d = os.path.join(ROOT, "graphsage_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/*_pmc_hbm.json: separate FETCH_SIZE and WRITE_SIZE runs of this same command).
    gfx950 correction: FETCH_SIZE counts 128-B read requests as 64 B, so it is doubled.
    Returns (bytes | None, file | None, stale): stale = the file was taken with other kernel
    sources than the ones this run's library was built from (or does not say)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc_hbm.json"))):
        best = path
    if best is None:
        return None, None, None
    try:
        z = json.load(open(best))
        fetch_kb, write_kb = z["FETCH_SIZE"][kernel]["mean_KB"], z["WRITE_SIZE"][kernel]["mean_KB"]
        stale = z.get("csrc_hash") != csrc_hash()
        return (2.0 * fetch_kb + write_kb) * 1024.0, os.path.relpath(best, ROOT), stale
    except (KeyError, ValueError, OSError):
        return None, None, None


# Random rows served per second by the fabric, whatever their size (32 / 64 bytes), allocation kind
# or cache policy: tools/micro/line_fetch_probe.hip, 200 M uniformly random rows of a 10 M-row
# table, 30 combinations, 51-53.5 G rows/s.  Every row costs one 128-byte request (RDREQ_SIZES in
# profiles/*_pmc_hbm.json: no 32- or 64-byte requests), so this, not the byte peak, is what the
# step's gathers run into.
FABRIC_REQUEST_CEILING = 53.0e9


def measured_requests(workload, kernel):
    """128-byte fabric read requests per launch of `kernel` (profiles/*_pmc_hbm.json, RDREQ_SIZES)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc_hbm.json"))):
        best = path
    if best is None:
        return None
    try:
        return float(json.load(open(best))["RDREQ_SIZES"][kernel]["requests"])
    except (KeyError, ValueError, OSError):
        return None


# ---------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args, hidden, out_dim, layers, K, inference=False):
    """The CPU oracle (oracle/sage_oracle.py: the op sequence PyG + loss.py execute on CPU,
    plus clip + Adam) on a bounded sample: the same generator at `--cpu-nodes` nodes with the
    pair batch scaled by the same factor, so pairs per node per step match the GPU workload."""
    import torch
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    from oracle import sage_oracle as O
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    n = args.cpu_nodes
    nodes_full = args.nodes or WORKLOADS[args.workload][0]
    ei = powerlaw_graph(n, WORKLOADS[args.workload][1], seed=42)
    x = memory_features(n, seed=42)
    torch.manual_seed(0)
    if inference:
        model = O.OracleGraphSAGE(7, hidden, out_dim, layers, 0.0).eval()
        with torch.no_grad():
            model(x, ei)                                             # warm-up
            steps, t0 = 0, time.perf_counter()
            while steps < args.cpu_steps:
                model(x, ei)
                steps += 1
                if time.perf_counter() - t0 > 20.0:
                    break
        dt = (time.perf_counter() - t0) / steps
        return {"value": n / dt, "unit": "nodes/s", "cores": threads, "kind": "port",
                "sample": f"{n}-node/{ei.size(1)}-edge graph from the same generator "
                          f"(1/{max(nodes_full // n, 1)} of the workload), same model, {steps} "
                          f"forwards, {dt * 1e3:.1f} ms each"}
    b = max(1024, int(args.pairs_per_gpu * n / nodes_full))
    model = O.OracleGraphSAGE(7, hidden, out_dim, layers, 0.3).train()
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
    out = {
        "value": b / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
        "sample": f"{n}-node/{ei.size(1)}-edge graph from the same generator (1/{max(nodes_full // n, 1)} "
                  f"of the workload), {b} pairs/step (same pairs per node), {steps} steps, "
                  f"{dt * 1e3:.1f} ms/step",
    }
    if args.workload == "c4" and not args.nodes:
        out["full_size_configs"] = cpu_full_size_configs(K)
    return out


def cpu_full_size_configs(K):
    """the CPU path on BASELINE configs 1 and 2 IN FULL (they fit the host) and on
    config 3 at a reduced step count, at the reference's own batch of 1024 pairs; bounded by time
    caps (5 s, 5 s, 20 s + one step)."""
    import numpy as np
    import torch
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    from oracle import sage_oracle as O
    res = []
    z = np.load(os.path.join(ROOT, "tests", "golden", "c1_graph.npz"))
    graphs = [("C1: the reference's default 500-node graph (generator seed 42)",
               torch.from_numpy(z["edge_index"]), torch.from_numpy(z["features"]).float(), 64, 200, 5.0),
              ("C2: 100k-node/~2M-edge power-law graph", powerlaw_graph(100_000, 20, seed=42),
               memory_features(100_000, seed=42), 64, 8, 5.0),
              # C3 at a reduced step count: no warm-up step, at most 2 steps, and the
              # second only if the first left time under the cap
              ("C3: 1M-node/~20M-edge power-law graph, SAGE 7->128->128 (reduced step count)",
               powerlaw_graph(1_000_000, 20, seed=42), memory_features(1_000_000, seed=42), 128, 2, 20.0)]
    threads = torch.get_num_threads()
    for name, ei, x, hidden, max_steps, cap_s in graphs:
        n = x.size(0)
        used = min(threads, 4) if n < 10_000 else threads    # a 500-node step gains nothing beyond
        torch.set_num_threads(used)
        torch.manual_seed(0)
        model = O.OracleGraphSAGE(7, hidden, hidden, 2, 0.3).train()
        opt = O.make_adam(model)
        batches = [cooccurrence_batch(ei, n, 1024, K, seed=s) for s in range(2)]
        if n < 500_000:
            O.train_step(model, opt, x, ei, *batches[0])
        steps, t0 = 0, time.perf_counter()
        while steps < max_steps and time.perf_counter() - t0 < cap_s:
            O.train_step(model, opt, x, ei, *batches[steps % 2])
            steps += 1
        dt = (time.perf_counter() - t0) / steps
        res.append({"config": name, "nodes": n, "edges": int(ei.size(1)), "pairs_per_step": 1024,
                    "ms_per_step": dt * 1e3, "pairs_per_s": 1024 / dt, "steps": steps,
                    "cores": used})
        del model, opt, ei, x
    torch.set_num_threads(threads)
    # C4 in full takes ~30 s a step and ~150 GB of host memory: measured once with
    # `bench.py --cpu-c4-one-step` on the GPU box's host and kept under profiles/, not repeated here
    rec = os.path.join(ROOT, "profiles", "r03_cpu_c4_one_step.json")
    if os.path.exists(rec):
        with open(rec) as f:
            r = json.loads(f.read().strip().splitlines()[-1])
        res.append({"config": "C4: 10M-node/~200M-edge power-law graph, one step at the bench's batch",
                    "nodes": r["nodes"], "edges": r["edges"], "pairs_per_step": r["pairs_per_step"],
                    "ms_per_step": r["s_per_step"] * 1e3, "pairs_per_s": r["value"], "steps": r["steps"],
                    "cores": r["cores"], "recorded": "profiles/r03_cpu_c4_one_step.json (not re-measured by this run)"})
    return res


def cpu_c4_one_step(args):
    """SURVEY 8(d) "C4 one step (expect minutes)": the CPU oracle's training step on the full
    10 M-node / 200 M-edge graph at the bench's batch (2^20 pairs, K = 5), once, no warm-up.  Far too
    long for the default run, so it has a flag of its own and its line is committed under profiles/.
    The oracle materialises PyG's [E, F] message tensors (51 GB at F = 64): expect ~150 GB of host memory."""
    import threading
    import torch
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
    from oracle import sage_oracle as O
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    nodes, deg = WORKLOADS["c4"][0], WORKLOADS["c4"][1]
    done = threading.Event()

    def heartbeat():                      # a line every 30 s: a silent minutes-long step looks hung
        t0 = time.perf_counter()
        while not done.wait(30.0):
            log(f"cpu C4 step: still running, {time.perf_counter() - t0:.0f} s")
    threading.Thread(target=heartbeat, daemon=True).start()
    t0 = time.perf_counter()
    ei = powerlaw_graph(nodes, deg, seed=42)
    x = memory_features(nodes, seed=42)
    batch = cooccurrence_batch(ei, nodes, args.pairs_per_gpu, args.negatives, seed=1001)
    log(f"cpu C4 step: graph and batch ready after {time.perf_counter() - t0:.0f} s ({ei.size(1)} edges)")
    torch.manual_seed(0)
    model = O.OracleGraphSAGE(7, 64, 64, 2, args.dropout).train()
    opt = O.make_adam(model)
    t1 = time.perf_counter()
    stats = O.train_step(model, opt, x, ei, *batch)
    dt = time.perf_counter() - t1
    done.set()
    loss = float(stats[0]) if isinstance(stats, (tuple, list)) else float(stats)
    print(json.dumps({"metric": "trained co-occurrence pairs/sec (CPU oracle, one full C4 step)",
                      "value": args.pairs_per_gpu / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
                      "nodes": nodes, "edges": int(ei.size(1)), "pairs_per_step": args.pairs_per_gpu,
                      "negatives": args.negatives, "s_per_step": dt, "steps": 1, "loss": loss,
                      "sample": "the whole workload: one step, no warm-up"}), flush=True)


# ------------------------------------------------------------------- launch-path self-test
def run_selftest(args, R):
    """GSAGE_BENCH_DIST_SELFTEST=1 (tests/test_bench_launch_cpu.py): the rank set-up, barrier,
    max-over-ranks timing and result line of the real bench with the GPU work left out, so the
    spawn / relay / exit-code logic can run on a CPU-only machine over gloo."""
    fail = os.environ.get("GSAGE_BENCH_SELFTEST_FAIL_RANK")
    if fail is not None and int(fail) == R.rank:
        print(f"[bench r{R.rank}] self-test: failing on purpose", file=sys.stderr, flush=True)
        sys.exit(7)
    R.sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + R.rank))
    R.sync_all()
    elapsed = R.max_over_ranks(time.perf_counter() - t0)
    seen = R.count_ranks()
    if R.rank == 0:
        if os.environ.get("GSAGE_BENCH_SELFTEST_NOISE") == "1":
            print("not the result line")
        print(json.dumps({"metric": "launch self-test (no GPU work)", "value": 0.0, "unit": "none",
                          "n_gpus": R.world, "ranks_seen": seen, "backend": R.backend,
                          "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "selftest": True}),
              flush=True)


# --------------------------------------------------------------- step anatomy (training)
def live_counts(torch, ei, batch, nodes):
    """(rows the batch names, edges whose TARGET is such a row): what the touched-row backward
    walks and what the masked transposed gather fetches.  One batch, outside the timed region."""
    t, c, neg = batch
    flag = torch.zeros(nodes, dtype=torch.bool, device=ei.device)
    flag[t] = True
    flag[c] = True
    flag[neg.reshape(-1)] = True
    rows = int(flag.sum().item())
    edges = int(flag[ei[1]].sum().item())
    return rows, edges


def step_anatomy(args, summ, N, E, F, D, B, K, R, Le):
    """One entry per bracketed C-ABI call of the step, largest share first.  useful_bytes = what
    the call's algorithm has to read and write (fp32 rows, int32 CSR, int64 batch ids), per launch:

      segment_gather_F*          4(N+1) + 4E + 4F.E + 4F.N                      
      segment_gather_F*_skip     4(N+1) + 4E + 4F.Le + 4F.R + 4F.N + N/8        (live edges / rows only)
      segment_gather_hidden1_F*  4E + 8.slots + 64E (128E: wide training) + 4F.N + pack 72N + 64N
      project_fwd_8xF            64N read, 4F.N + F.N/8 (sign bits) written
      project_fwd_FxD            8F.N read, 4D.N + 4N written
      project_bwd_rows_FxD       R rows of grad, out, agg, x read; R rows of d agg, d x written
      project_bwd_8xF_weights_only   4F.N (gradient) + F.N/8 (sign bits) + 64N
      hidden1_wgrad_F*           68.Le (id + 64-byte row per live edge) + slot list + R rows of d agg, d x
      skipgram_fwd               (2+K)(4D + 8).B                                 
      skipgram_bwd_*             (2+K).4D.B re-gathered + 2(2+K).4D.B read-modify-write
      touched_rows / zero_rows   8(2+K).B ids + N flag bytes + N/8 mask + 4R list / 4D.R + 4R
    """
    train = args.dropout > 0
    wide_train = train and F > 64
    slots = N + E // 8                      # bound of the slot list (hidden_gather.hip)
    models = {
        "segment_gather_F8": (["seg_gather_vec_k<2, 0, false>"],
                              "layer-1 mean aggregation of the padded input features (32-byte rows)",
                              4 * (N + 1) + 4 * E + 32 * E + 32 * N),
        f"segment_gather_F{F}": ([f"seg_gather_vec_k<{F // 4}, 0, false>"],
                                 f"mean aggregation over stored {4 * F}-byte rows",
                                 4 * (N + 1) + 4 * E + 4 * F * E + 4 * F * N),
        f"segment_gather_F{F}_skip": ([f"seg_gather_vec_k<{F // 4}, 0, true>"],
                                      "transposed (backward) gather of the layer-2 aggregation; rows "
                                      "whose gradient is zero are not fetched",
                                      4 * (N + 1) + 4 * E + 4 * F * Le + 4 * F * R + 4 * F * N + N // 8),
        f"segment_gather_hidden1_F{F}": ([f"hidden1_gather_k<{F // 32}, {'true' if train else 'false'}>",
                                          "h1_pack_u_k", "h1_pack_w_k", "h1_hub_combine_k"],
                                         "layer-2 forward aggregation recomputed from the input layer's "
                                         "64-byte operand rows on the matrix cores",
                                         4 * E + 8 * slots + (128 if wide_train else 64) * E + 4 * F * N
                                         + 72 * N + 64 * N),
        f"project_fwd_8x{F}": (["project_fwd_k16_k", "sign_flag_k"],
                               "layer-1 projection + ReLU + dropout, sign bits for the backward",
                               64 * N + 4 * F * N + F * N // 8),
        f"project_fwd_{F}x{D}": (["proj64h_fwd_k", "p64h_pack_k", "proj64_fwd_k", "pack_fwd_k"]
                                 if F == 64 and D == 64 else
                                 [f"gemm_split_k<{D // 64}>", "split_planes_k"],
                                 "layer-2 projection + L2 norm", 8 * F * N + 4 * D * N + 4 * N),
        f"project_bwd_rows_{F}x{D}": ([{"f": "bwd64_fused_k<2, true, false>", "d": "bwd64_fused_k<2, true, true>"}.get(
                                           os.environ.get("GSAGE_BWD64", "")[:1], "bwd64h_rows_k<2>"),
                                       "reduce_slabs_k", "pack_dx_k", "bwd64h_pack_k"],
                                      "fused layer-2 backward (L2-norm', dX, dW) over the touched rows",
                                      R * (4 * D * 2 + 4 * F * 2) + R * 8 * F + 4 * R),
        f"project_bwd_8x{F}_weights_only": (["dw16_mfma_k", "reduce_dw16_k"],
                                            "layer-1 weight gradients (ReLU'/dropout' from the sign bits)",
                                            4 * F * N + F * N // 8 + 64 * N),
        f"hidden1_wgrad_F{F}": (["hidden1_wgrad_k", "wg_fill_k", "wg_count_k", "wg_reduce1_k"],
                                "layer-1 weight gradients from layer 2's backward: one 64-byte operand row "
                                "per live edge on the matrix cores (replaces the transposed gather + dW pass)",
                                # per live edge its id and the 64-byte row; the slot list written and read
                                # (ceil(deg / 8) slots per touched row); d agg, d x, the own 64-byte row
                                # (root term) and the list entry of every touched row; the waves' partial sums
                                (4 + 64) * Le + 16 * (R + Le // 8) + R * (8 * F + 4 + 8 + 64) + 2 * 6144 * 4096),
        "skipgram_fwd": (["skipgram_fwd_vec_k<1>", "skipgram_finish_k"], "skip-gram loss forward",
                         (2 + K) * (4 * D + 8) * B),
        "skipgram_bwd_atomic": (["skipgram_bwd_k"], "skip-gram loss backward (float atomics)",
                                3 * (2 + K) * 4 * D * B),
        "skipgram_bwd_det": ([f"seg_gather_vec_k<{D // 4}, 2, false>", "radix_scatter_k<true>*3", "sg_terms_k"],
                             "skip-gram loss backward (sorted, deterministic)", 3 * (2 + K) * 4 * D * B),
        "skipgram_bwd_rows": (["sg_rows_gather_k<16>", "radix_scatter_k<true>*3", "sg_terms_rows_k"],
                              # (radix_hist_k and rowptr_from_sorted_k also run in the CSR build, on 200 M
                              # keys: their per-launch means in the counter file are not this call's)
                              "skip-gram loss backward: terms sorted by place in the touched-row list, "
                              "summed row by row (deterministic, no atomics)",
                              # terms listed (ids read, 12 B written), three radix passes over (row, source,
                              # weight): 4 B for the histogram, 12 B read and 12 B written per pass; row
                              # boundaries; then the sums: one source row per term, one written row per
                              # listed row
                              2 * (1 + K) * B * (16 + 12 + 3 * 28 + 4 + 8 + 4 * D) + R * (4 * D + 8)),
        "touched_rows": (["rows_mark_k", "rows_count_k", "rows_scan_k", "rows_emit_k"],
                         "which rows the batch names (mask + sorted list)",
                         8 * (2 + K) * B + 2 * N + N // 8 + 4 * R),
        "zero_rows": (["rows_zero_k"], "clear the touched rows of the gradient table", 4 * D * R + 4 * R),
    }
    out = []
    for tag, (launches, ms) in summ.items():
        kernels, what, useful = models.get(tag, ([], tag, None))
        traffic, src, stale = None, None, None
        if not args.nodes and kernels:
            # "name*3": that kernel runs three times per call (the file holds means per launch)
            parts = []
            for k in kernels:
                kname, _, times = k.partition("*")
                t_, src_, stale_ = measured_traffic(args.workload, kname)
                parts.append((t_ * int(times or 1) if t_ is not None else None, src_, stale_))
            got = [p for p in parts if p[0] is not None]
            if got and parts[0][0] is not None:          # the main kernel must be in the file
                traffic = sum(p[0] for p in got)
                src, stale = got[0][1], any(p[2] for p in got)
        sec = ms * 1e-3
        entry = {"tag": tag, "kernels": kernels, "what": what, "launches": launches, "ms": ms,
                 "useful_bytes": useful,
                 "useful_GBs": useful / sec / 1e9 if useful else None,
                 "frac_useful": useful / sec / 1e9 / HBM_PEAK_GBS if useful else None,
                 "traffic": traffic, "traffic_source": src, "traffic_stale": stale,
                 "frac_traffic": traffic / sec / 1e9 / HBM_PEAK_GBS if traffic else None}
        if not args.nodes and kernels and (tag.startswith("segment_gather") or tag.startswith("hidden1_wgrad")):
            # the gathers: bound by the fabric's request rate (one 128-byte request per row), not by bytes
            req = measured_requests(args.workload, kernels[0].partition("*")[0])
            if req:
                entry.update(fabric_requests=req, Grequests_per_s=req / sec / 1e9,
                             frac_of_request_ceiling=req / sec / FABRIC_REQUEST_CEILING)
        out.append(entry)
    out.sort(key=lambda a: -a["ms"] * a["launches"])
    return out


# ------------------------------------------------------------------------------- training
def run_training(args, R):
    torch = R.torch
    from graphsage_amd import GraphSAGE, TrainStep, default_cache, ops
    from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph

    nodes, deg, hidden, out_dim, layers, desc = WORKLOADS[args.workload]
    if args.nodes:
        nodes, desc = args.nodes, desc + f" [node count overridden: {args.nodes}]"
    K, B = args.negatives, args.pairs_per_gpu
    dev, rank, world = R.dev, R.rank, R.world

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
    model = GraphSAGE(7, hidden, out_dim, layers, args.dropout,
                      cache_input_aggregation=args.cache_input_aggregation).to(dev)
    step = TrainStep(model, x, ei, receptive_field=args.receptive_field)
    nbatch = 4
    batches = [cooccurrence_batch(ei, nodes, B, K, seed=1000 * (rank + 1) + i) for i in range(nbatch)]
    torch.cuda.synchronize()

    for i in range(args.warmup):
        step.step(*batches[i % nbatch], global_batch=B * world)
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    R.sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        stats = step.step(*batches[i % nbatch], global_batch=B * world)
    R.sync_all()
    elapsed = R.max_over_ranks(time.perf_counter() - t0)
    ops.set_kernel_timer(None)
    loss = float(stats[0].item())
    seen = R.count_ranks()

    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    # ---- step anatomy and roofline.  Every C-ABI call of the step is bracketed by HIP events on
    # the launch stream (ops.KernelTimer).  Per bracket: live mean duration, the bytes it MUST move
    # for the algorithm it runs (`useful_bytes`, formulas below: DESIGN.md section 3), the HBM bytes
    # the counters saw for its kernels (`traffic`, profiles/*_pmc_hbm.json: 2 x FETCH_SIZE +
    # WRITE_SIZE, separate passes), and both as fractions of the 8 TB/s peak.  `roofline` is the
    # bracket with the largest share of the step; `roofline.survey_model` keeps SURVEY 8(d)'s
    # figure for the layer-2 forward aggregation (the bytes of a gather of STORED 256-byte rows),
    # which the recompute gather does not move, as an equivalent rate and a speed-up.
    F = hidden
    summ = timer.summary()
    live_rows, live_edges = live_counts(torch, ei, batches[0], nodes)
    anatomy = step_anatomy(args, summ, nodes, E, F, out_dim, B, K, live_rows, live_edges)
    alg_bytes = 4 * (nodes + 1) + 4 * E + 4 * E * F + 4 * nodes * F     # SURVEY 8(d), layer-2 fwd
    rec = next((a for a in anatomy if a["tag"] == f"segment_gather_hidden1_F{F}"), None)
    plain = next((a for a in anatomy if a["tag"] == f"segment_gather_F{F}"), None)
    top = anatomy[0] if anatomy else None
    roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "csrc_hash": csrc_hash()}
    if top is not None:
        roofline.update({
            "kernel": top["kernels"][0] if top["kernels"] else top["tag"], "what": top["what"],
            "tag": top["tag"], "launches_timed": top["launches"], "mean_ms": top["ms"],
            # achieved = algorithmic bytes of THIS kernel's algorithm per launch / its duration
            "achieved": top["useful_GBs"], "frac": top["frac_useful"],
            "useful_bytes": top["useful_bytes"], "frac_useful": top["frac_useful"],
            "traffic": top["traffic"], "frac_traffic": top["frac_traffic"],
            "traffic_source": top["traffic_source"], "traffic_stale": top["traffic_stale"],
            "share_of_step": top["ms"] * top["launches"] / args.steps / ms_per_step,
            "note": "dominant bracket of the step by live HIP-event time; frac = useful bytes / time / "
                    "peak (never above 1), frac_traffic = counter bytes / time / peak; for the gathers "
                    "frac_of_request_ceiling = 128-byte fabric requests / time over the 53 G random "
                    "rows/s the fabric serves whatever the row size (tools/micro/line_fetch_probe.hip)"})
        for k in ("fabric_requests", "Grequests_per_s", "frac_of_request_ceiling"):
            if k in top:
                roofline[k] = top[k]
    agg_l2 = rec or plain
    if agg_l2 is not None:
        sm = {"what": "SURVEY 8(d): bytes of the layer-2 forward aggregation as a gather of stored rows",
              "bytes": alg_bytes, "kernel": agg_l2["kernels"][0], "mean_ms": agg_l2["ms"],
              "equivalent_GBs": alg_bytes / (agg_l2["ms"] * 1e-3) / 1e9}
        if rec is not None and world == 1 and not args.nodes:
            # the kernel it replaces, on the same graph: three launches outside the timed steps
            ops.set_kernel_timer(None)
            h = torch.rand(nodes, F, device=dev)
            ops.aggregate_mean(graph, h)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.aggregate_mean(graph, h)
            e1.record()
            torch.cuda.synchronize()
            pms = e0.elapsed_time(e1) / 3
            del h
            plain_name = f"seg_gather_vec_k<{F // 4}, 0, false>"
            p_traffic, _, _ = measured_traffic(args.workload, plain_name)
            sm["plain_gather"] = {
                "kernel": f"{plain_name}: the same aggregation over the stored activation "
                          "(GSAGE_HIDDEN1_RECOMPUTE=0), 3 launches outside the timed steps",
                "mean_ms": pms, "useful_bytes": alg_bytes,
                "frac_useful": alg_bytes / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": p_traffic,
                "frac_traffic": p_traffic / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS if p_traffic else None}
            sm["speedup_vs_plain_gather"] = pms / agg_l2["ms"]
        roofline["survey_model"] = sm
    roofline["kernels"] = anatomy[:8]
    roofline["bracketed_ms_per_step"] = sum(a["ms"] * a["launches"] for a in anatomy) / args.steps

    # the batch the reference would actually train with (config/config.yaml:37, trainer.py:108):
    # a short second timed loop on the same graph, model and harness, eager
    ref_batch = None
    if world == 1 and not args.no_reference_batch:
        ref_batch = reference_batch_leg(args, R, step, ei, nodes, K)

    # fp32 storage and accumulation everywhere; the 128/256-wide GEMMs (C3) feed the bf16 matrix
    # pipe with each fp32 operand split exactly into three bf16 parts (fp32-equivalent results)
    dtype = "f32" if hidden < 128 or os.environ.get("GSAGE_WIDE_GEMM", "")[:1] == "f" else \
        "f32 (wide GEMMs: bf16 MFMA on 3-way split fp32 operands, fp32 accumulate)"
    # two kernels feed the fp16 matrix pipe with each fp32 operand split into two fp16 parts, scaled by
    # powers of two taken from max |operand| (three of the four products, fp32 accumulate; error
    # <= ~1e-6 of the largest pre-activation, inside the 2e-6 parity bar): the recomputed layer-2
    # aggregation and, when the range of the hidden activations is handed over, the 64 -> 64 forward
    halves = []
    if any(a["tag"].startswith("segment_gather_hidden1") for a in anatomy):
        halves.append("the layer-2 forward aggregation (recomputes the 16->hidden projection of every "
                      "gathered row)")
    from graphsage_amd import ops as _ops
    if hidden == 64 and out_dim == 64 and _ops.RANGE_WORDS and os.environ.get("GSAGE_PROJ64", "")[:1] != "f":
        halves.append("the 64->64 forward projection")
    if any(a["tag"].startswith("project_bwd_rows") for a in anatomy) and os.environ.get("GSAGE_BWD64", "")[:1] != "f":
        halves.append("the dX half of the fused 64->64 backward" if os.environ.get("GSAGE_BWD64", "")[:1] == "d"
                      else "the fused 64->64 backward over the touched rows (dX and dW)")
    if any(a["tag"].startswith("hidden1_wgrad") for a in anatomy):
        halves.append("the input layer's weight gradients (one pass over the live edges)")
    if halves:
        dtype += ("; " if dtype != "f32" else " (storage, accumulation and every other GEMM in fp32), except: ") + \
            ", ".join(halves) + (" run on fp16 MFMA with 2-way split fp32 operands scaled by powers of "
                                 "two, 3 of 4 products, fp32 accumulate: fp32-equivalent to ~1e-6 of "
                                 "the largest pre-activation / gradient")
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle on the bounded sample ...")
            cpu = cpu_baseline(args, hidden, out_dim, layers, K)
        print(json.dumps({
            "metric": "co-occurrence pairs/sec trained (fwd+bwd+step)", "value": value,
            "unit": "pairs/s", "n_gpus": world, "ranks_seen": seen, "backend": R.backend,
            "steps": args.steps, "warmup": args.warmup,
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
                       "loss_backward": "deterministic (sorted over the touched rows)"
                       if ops.loss_bwd_rowwise(nodes, out_dim, B, K) else
                       "deterministic (sorted)" if ops.loss_bwd_deterministic(nodes, B, K) else "atomic",
                       "final_loss": loss},
            "roofline": roofline, "reference_batch": ref_batch, "cpu_baseline": cpu,
        }), flush=True)


def reference_batch_leg(args, R, step, ei, nodes, K, pairs=1024, steps=10, warmup=2):
    torch = R.torch
    from graphsage_amd.synth import cooccurrence_batch
    batches = [cooccurrence_batch(ei, nodes, pairs, K, seed=77 + i) for i in range(2)]

    def timed(n_warm, n_steps):
        for i in range(n_warm):
            step.step(*batches[i % 2], global_batch=pairs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step.step(*batches[i % 2], global_batch=pairs)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_steps

    was = step.receptive_field
    step.receptive_field = False
    dt = timed(warmup, steps)
    out = {"pairs": pairs, "ms_per_step": dt * 1e3, "pairs_per_s": pairs / dt, "steps": steps,
           "graph_pass": "full graph every step (as the reference), eager",
           "source": "config/config.yaml:37 batch_size=1024"}
    # the same batch through the opt-in 2-hop-closure step (same parameters afterwards): labelled
    step.receptive_field = True
    dt_rf = timed(warmup, steps)
    out["receptive_field_opt_in"] = {"ms_per_step": dt_rf * 1e3, "pairs_per_s": pairs / dt_rf,
                                     "graph_pass": "2-hop closure of the batch only (opt-in)"}
    step.receptive_field = was
    return out


# ------------------------------------------------------------------------------ inference
def run_inference(args, R):
    """C5: the device part of StructuralEncoder.encode_all (encoder.py:181-185) timed as
    scripts/benchmark_latency.py:66-105 does: synchronise, K forwards one by one, synchronise."""
    torch = R.torch
    from graphsage_amd import GraphSAGE, default_cache, ops
    from graphsage_amd.encoder import StructuralEncoder
    from graphsage_amd.synth import memory_features, powerlaw_graph

    nodes, deg, hidden, out_dim, layers, desc = WORKLOADS[args.workload]
    if args.nodes:
        nodes, desc = args.nodes, desc + f" [node count overridden: {args.nodes}]"
    dev, rank, world = R.dev, R.rank, R.world
    t0 = time.perf_counter()
    ei = powerlaw_graph(nodes, deg, seed=42, device=dev)
    x = memory_features(nodes, seed=42, device=dev)
    E = ei.size(1)
    graph = default_cache().get(ei, nodes)
    torch.cuda.synchronize()
    log(f"graph: N={nodes} E={E} generated + CSR in {time.perf_counter() - t0:.1f}s")
    torch.manual_seed(0)
    enc = StructuralEncoder(model=GraphSAGE(7, hidden, out_dim, layers, 0.0), device=str(dev))
    for _ in range(args.warmup):
        emb = enc.encode_tensors(x, ei)
    del graph
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    R.sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        emb = enc.encode_tensors(x, ei)
    R.sync_all()
    elapsed = R.max_over_ranks(time.perf_counter() - t0)
    ops.set_kernel_timer(None)
    seen = R.count_ranks()
    norms = emb.norm(dim=1)
    unit = bool(((norms - 1).abs() < 1e-4).all().item())
    ms = elapsed / args.steps * 1e3
    summ = timer.summary()

    # the 256 -> 256 projections (layers 2 and 3): [agg|x] (N x 512) . [W_l|W_r]^T (512 x 256) on
    # the 16-bit matrix pipe with every fp32 operand split exactly: gemm_half_k<4>
    # (two fp16 parts, three products per fp32 product) for 256-column tiles with K >= 256,
    # gemm_split_k (three bf16 parts, six products) otherwise; GSAGE_WIDE_GEMM forces a family
    forced = os.environ.get("GSAGE_WIDE_GEMM", "")[:1]
    family = {"f": "f32", "s": "bf16x3", "h": "fp16x2"}.get(
        forced, "fp16x2" if out_dim == 256 and 2 * hidden >= 256 else "bf16x3")
    products = {"f32": 1.0, "bf16x3": 6.0, "fp16x2": 3.0}[family]
    kname = {"f32": "gemm_wide_k<2>", "bf16x3": "gemm_split_k<4>", "fp16x2": "gemm_half_k<4>"}[family]
    launches, mean_ms = summ.get(f"project_fwd_{hidden}x{out_dim}", (0, float("nan")))
    f32_flops = 2.0 * nodes * (2 * hidden) * out_dim
    issued = f32_flops * products
    peak = MFMA_F32_PEAK_TFS if family == "f32" else MFMA_BF16_PEAK_TFS
    ach = issued / (mean_ms * 1e-3) / 1e12 if launches else None
    # its streams: agg and x read once, out written once, fp32 (the 0.5 MB of weights stay in L2)
    gemm_bytes = 4.0 * nodes * (2 * hidden + out_dim)
    gbs = gemm_bytes / (mean_ms * 1e-3) / 1e9 if launches else None
    mfma = {"achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak if launches else None,
            "flops_counted": f"MFMA flops issued = {products:g} products per fp32 product",
            "fp32_equivalent_tflops": f32_flops / (mean_ms * 1e-3) / 1e12 if launches else None,
            "fp32_equivalent_frac_of_f32_matrix_peak":
                f32_flops / (mean_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFS if launches else None,
            "algorithmic_flops": f32_flops}
    hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": gbs / HBM_PEAK_GBS if launches else None, "algorithmic_bytes": gemm_bytes}
    # the bound is the larger of the two floors (flops / matrix peak against bytes / HBM peak): the
    # fp32 and bf16x3 kernels are matrix-bound; with three products the streams weigh more
    mfma_floor = issued / (peak * 1e12)
    hbm_floor = gemm_bytes / (HBM_PEAK_GBS * 1e9)
    bound = "mfma" if mfma_floor >= hbm_floor else "hbm"
    proj = dict(mfma if bound == "mfma" else hbm)
    proj.update({
        "bound": bound,
        "kernel": f"{kname} ({hidden}->{out_dim} projection; the bracket also holds the weight "
                  "pack/split, microseconds)",
        "floors_ms": {"mfma": mfma_floor * 1e3, "hbm": hbm_floor * 1e3},
        "other_bound": hbm if bound == "mfma" else mfma,
        "launches_timed": launches, "mean_ms": mean_ms if launches else None})
    traffic, traffic_src, stale = measured_traffic("c5", kname) if not args.nodes else (None, None, None)
    proj.update({"traffic": traffic, "traffic_source": traffic_src, "traffic_stale": stale,
                 "frac_traffic": traffic / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                 if traffic and launches else None})
    dtype = {"f32": "f32",
             "bf16x3": "f32 (wide GEMMs: bf16 MFMA on 3-way split fp32 operands, fp32 accumulate)",
             "fp16x2": "f32 (wide GEMMs: fp16 MFMA on 2-way split, range-scaled fp32 operands, fp32 "
                       "accumulate; tiles outside the range on bf16 3-way split)"}[family]
    F = hidden
    alg_bytes = 4 * (nodes + 1) + 4 * E + 4 * E * F + 4 * nodes * F        # SURVEY 8(d), per layer
    blocked_tag = f"segment_gather_F64x{F // 64}_blocked"
    per_launch = 1
    if blocked_tag in summ:
        # hidden activations kept as F/64 tables of [N,64]: one layer's aggregation = F/64 launches
        # of the 64-wide gather (bracketed together); priced on the SAME algorithmic bytes as the
        # single 256-wide gather it replaces (the col re-reads of launches 2..4 are not counted)
        gl, gms = summ[blocked_tag]
        per_launch = F // 64
        gkernel = "seg_gather_vec_k<16, 0, false>"
        gname = f"{F // 64} x {gkernel} over column-blocked [N,64] tables, F={F}"
    else:
        gl, gms = summ.get(f"segment_gather_F{F}", (0, float("nan")))
        gkernel = f"seg_gather_vec_k<{F // 4}, 0, false>"
        gname = f"{gkernel}, F={F}"
    gach = alg_bytes / (gms * 1e-3) / 1e9 if gl else None
    g_traffic, g_src, g_stale = measured_traffic("c5", gkernel) if not args.nodes else (None, None, None)
    if g_traffic is not None:
        g_traffic *= per_launch                      # the file holds the mean per kernel launch
    gather = {"bound": "hbm", "kernel": gname, "achieved": gach, "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": gach / HBM_PEAK_GBS if gl else None,
              "useful_bytes": alg_bytes, "frac_useful": gach / HBM_PEAK_GBS if gl else None,
              "traffic": g_traffic, "traffic_source": g_src, "traffic_stale": g_stale,
              "frac_traffic": g_traffic / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS if g_traffic and gl else None,
              "launches_timed": gl, "mean_ms": gms if gl else None,
              "what": "mean aggregation of a 256-wide hidden layer over stored rows (SURVEY 8d bytes)"}
    # the dominant bracket of the encode by live time is named `roofline`; the other one rides along
    gather_share = gl * gms if gl else 0.0
    proj_share = launches * mean_ms if launches else 0.0
    if gather_share >= proj_share:
        roofline = dict(gather)
        roofline["projection"] = proj
    else:
        roofline = dict(proj)
        roofline["gather"] = gather
    roofline["share_of_step"] = max(gather_share, proj_share) / args.steps / ms
    roofline["csrc_hash"] = csrc_hash()
    htag = f"segment_gather_hidden1_F{F}"
    if htag in summ:
        # the first wide aggregation (layer 2) is recomputed from the input layer's operands; its
        # own bytes: col + slot descriptors + one 64-byte row per neighbour + the output + the pack
        hl, hms = summ[htag]
        hname = f"hidden1_gather_k<{F // 32}, false>"
        h_traffic = None
        if not args.nodes:
            parts = [measured_traffic("c5", k)[0] for k in (hname, "h1_pack_u_k", "h1_hub_combine_k")]
            h_traffic = sum(p for p in parts if p) if parts[0] else None
        h_useful = 4 * E + 8 * (nodes + E // 8) + 64 * E + 4 * F * nodes + 136 * nodes
        roofline["gather_recomputed"] = {
            "bound": "hbm", "kernel": f"{hname} + pack kernels, F={F} (layer 2)", "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "launches_timed": hl, "mean_ms": hms,
            "useful_bytes": h_useful, "achieved": h_useful / (hms * 1e-3) / 1e9,
            "frac": h_useful / (hms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_useful": h_useful / (hms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": h_traffic,
            "frac_traffic": h_traffic / (hms * 1e-3) / 1e9 / HBM_PEAK_GBS if h_traffic else None,
            "survey_model": {"bytes": alg_bytes, "equivalent_GBs": alg_bytes / (hms * 1e-3) / 1e9,
                             "speedup_vs_plain_gather": gms / hms if gl else None,
                             "what": "SURVEY 8(d) bytes of the gather of stored 1 KiB rows it stands in for"},
            "note": "fetches 64 bytes per neighbour instead of 1 KiB and recomputes the rows on the matrix "
                    "cores; limited by instruction issue (VALU + MFMA), then by the 128-byte line rate"}
    per_kernel = {k: {"launches": v[0], "mean_ms": v[1]} for k, v in sorted(summ.items())}
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle forward on the bounded sample ...")
            cpu = cpu_baseline(args, hidden, out_dim, layers, 0, inference=True)
        print(json.dumps({
            "metric": "full-graph inference encode, nodes/sec (BASELINE config 5)",
            "value": nodes * world * args.steps / elapsed, "unit": "nodes/s", "n_gpus": world,
            "ranks_seen": seen, "backend": R.backend, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc, "nodes": nodes, "edges": E,
                       "layers": f"7->{hidden}->{hidden}->{out_dim}, eval, no_grad, L2-normalised",
                       "parallelism": "replicas only (inference does not shard)" if world > 1
                       else "single GPU", "unit_norm_rows": unit,
                       "protocol": "scripts/benchmark_latency.py:66-105"},
            "roofline": roofline, "kernels_ms": per_kernel, "cpu_baseline": cpu,
        }), flush=True)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.cpu_c4_one_step:
        cpu_c4_one_step(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))       # before any GPU call in this process
    R = Ranks(args)
    if SELFTEST:
        run_selftest(args, R)
    elif args.workload == "c5":
        run_inference(args, R)
    else:
        run_training(args, R)
    R.finish()


if __name__ == "__main__":
    main()
