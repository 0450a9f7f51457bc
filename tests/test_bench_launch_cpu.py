"""`python bench.py --gpus N` without a launcher must start N ranks itself, relay rank 0's one
JSON line and fail loudly otherwise.  Exercised here on the CPU over gloo
with GSAGE_BENCH_DIST_SELFTEST=1, which keeps bench.py's own launch / rank set-up / barrier /
max-over-ranks / result-line code and leaves the GPU work out."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **extra_env):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GSAGE_BENCH_DIST_SELFTEST="1", **extra_env)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True,
                          timeout=240)


def _result_lines(stdout):
    return [ln for ln in stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(300)
def test_gpus_2_spawns_two_ranks_and_relays_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0"], GSAGE_BENCH_SELFTEST_NOISE="1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _result_lines(r.stdout)
    assert len(lines) == 1 and r.stdout.strip() == lines[0]     # nothing but the line on stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["backend"] == "gloo"
    assert rec["steps"] == 3
    # max over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1 ms
    assert rec["ms_per_step"] >= 2.0


@pytest.mark.timeout(300)
def test_failing_rank_gives_nonzero_exit_and_no_line():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], GSAGE_BENCH_SELFTEST_FAIL_RANK="1")
    assert r.returncode != 0
    assert _result_lines(r.stdout) == []
    assert "FAILED" in r.stderr


@pytest.mark.timeout(120)
def test_gpus_must_match_world_size():
    # a launcher gave this process WORLD_SIZE=1 semantics (unset + --gpus 1 is the plain case);
    # WORLD_SIZE=1 with --gpus 2 must refuse to print a line
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    env.update(GSAGE_BENCH_DIST_SELFTEST="1", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=100)
    assert r.returncode != 0
    assert _result_lines(r.stdout) == []
    assert "refusing" in r.stderr


@pytest.mark.timeout(120)
def test_single_rank_line_reports_one_rank():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(_result_lines(r.stdout)[0])
    assert rec["n_gpus"] == 1 and rec["ranks_seen"] == 1
