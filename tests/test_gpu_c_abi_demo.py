"""GPU: the torch-free C++ host program (examples/c_abi_demo.cpp, built by
__graft_entry__.build()) reproduces the hand-computed known answers KAT-1 / KAT-5 through the
C ABI alone: the drop-in boundary does not depend on Python or torch."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_program_reproduces_known_answers():
    exe = os.path.join(ROOT, "examples", "c_abi_demo")
    if not os.path.exists(exe):
        pytest.skip("examples/c_abi_demo not built: run __graft_entry__.build() (hipcc) first")
    # a child process of its own: nothing is exec'd from this (GPU-initialised) interpreter
    try:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    except OSError as e:          # a sandbox that forbids starting programs from here
        pytest.skip(f"cannot start a child process in this environment: {e}")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all known answers reproduced through the C ABI" in r.stdout
    assert "MISMATCH" not in r.stdout
