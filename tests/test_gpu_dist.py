"""GPU: the multi-rank step on real kernels.  Two ranks share the one GPU of the box over gloo
(RCCL needs a GPU per rank): pair shards + one all-reduce of [gradients | stats] must reproduce
the single-process step on the whole batch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_match_single_process():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29611",
           os.path.join(ROOT, "tools", "dist_gpu_check.py")]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
    except OSError as e:
        pytest.skip(f"cannot start child processes in this environment: {e}")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world"] == 2 and d["dropout"] == 0.3     # the reference's rate (graphsage.py:135)
    for a, b in zip(d["losses"], d["ref_losses"]):
        assert a == pytest.approx(b, rel=1e-5)
    assert d["max_param_diff"] <= 1e-5
    assert d["live_rows_used"] == 6          # 3 sharded steps + 3 whole-batch steps on rank 0
    # the captured form (two HIP graphs per rank around the all-reduce, masks from the stepper's
    # own graph-safe generator seeded with the shared base seed) against one captured process
    for a, b in zip(d["captured_losses"], d["captured_ref_losses"]):
        assert a == pytest.approx(b, rel=1e-5)
    assert d["captured_max_param_diff"] <= 1e-5
