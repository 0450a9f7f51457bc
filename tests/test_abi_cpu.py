"""CPU: the C-ABI library loads, exports every symbol include/gsage.h declares, and the
host-only entry points behave; the product path refuses CPU tensors (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gsage.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsage_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from graphsage_amd import _lib
    names = _declared()
    assert len(names) >= 13
    lib = ctypes.CDLL(_lib.lib_path())
    for n in names:
        assert hasattr(lib, n), f"libgsage.so does not export {n}"
    assert sorted(_lib.SIGNATURES) == names        # python binding covers the header 1:1


def test_host_only_entry_points():
    from graphsage_amd import _lib
    lib = _lib.load()
    assert lib.gsage_version() == 100
    assert lib.gsage_error_string(0) == b"ok"
    assert b"workspace" in lib.gsage_error_string(-3)
    assert lib.gsage_long_row_threshold() > 64
    assert lib.gsage_csr_workspace_bytes(0, 0) > 0
    small, big = lib.gsage_csr_workspace_bytes(1000, 100), lib.gsage_csr_workspace_bytes(10**6, 10**5)
    assert big > small >= 6 * 4 * 1000
    assert lib.gsage_project_workspace_bytes(64, 64) >= 2 * 64 * 64 * 4
    assert lib.gsage_project_bwd_workspace_bytes(10**6, 64, 64) > lib.gsage_project_workspace_bytes(64, 64)
    assert lib.gsage_skipgram_workspace_bytes(1024) > 0
    assert lib.gsage_csr_workspace_bytes(-1, 5) == 0


def test_argument_validation_without_gpu():
    """Entry points check their arguments before touching the device."""
    from graphsage_amd import _lib
    lib = _lib.load()
    assert lib.gsage_segment_gather(None, None, None, None, None, None, 10, 64, None, None, None, None) == -1
    assert lib.gsage_csr_build(None, 5, 5, None, None, None, None, None, None, None, 0, None, None, None, 0, None) == -1
    assert lib.gsage_project_fwd(None, None, None, None, None, 4, 7, 64, 0, None, 0, 0.0, None, None, None, 0, None) == -1
    assert lib.gsage_skipgram_fwd(None, 4, 64, None, None, None, 8, 5, 5.0, 8.0, 40.0, None, None, None, 0, None) == -1
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(RuntimeError):
        _lib.check(-4, "x")


def test_product_path_has_no_cpu_fallback():
    from graphsage_amd import GraphSAGE, SkipGramLoss
    m = GraphSAGE()
    with pytest.raises(ValueError, match="GPU"):
        m(torch.randn(10, 7), torch.randint(0, 10, (2, 30)))
    with pytest.raises(ValueError, match="GPU|CPU"):
        SkipGramLoss()(torch.randn(10, 64), torch.tensor([1]), torch.tensor([2]), torch.tensor([[3]]))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "graphsage_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read(), f


def test_round2_entry_points_validate_before_touching_the_device():
    from graphsage_amd import _lib
    lib = _lib.load()
    # touched rows / zero rows
    assert lib.gsage_touched_rows_workspace_bytes(0) == 0
    assert lib.gsage_touched_rows_workspace_bytes(10_000_000) >= 10_000_000      # a flag byte per row
    assert lib.gsage_touched_rows(None, None, None, 8, 5, 100, None, None, 10, None, None, 0, None) == -1
    assert lib.gsage_zero_rows(None, None, 10, 64, None, None) == -1
    # the row-list backward exists for 64 -> 64 only; other shapes say so instead of guessing
    assert lib.gsage_project_bwd_rows_supported(64, 64) == 1
    assert lib.gsage_project_bwd_rows_supported(128, 128) == 0 and lib.gsage_project_bwd_rows_supported(8, 64) == 0
    assert lib.gsage_project_bwd_rows(None, None, None, None, None, None, None, None, 10, 128, 128, 3, 0.0,
                                      None, None, None, None, None, None, 0, None, None, 0, None) == -2
    # the add_to mask is not a hint: widths that cannot honour it are refused
    assert lib.gsage_segment_gather_live(None, None, None, None, None, None, 10, 64, None, None, None,
                                         None, None) == -1
    # column-blocked projection: which shapes have a kernel
    sup = lib.gsage_project_fwd_blocked_supported
    assert sup(8, 256, 0, 1, _lib.EPI_RELU) == 1 and sup(256, 256, 1, 1, _lib.EPI_RELU) == 1
    assert sup(256, 256, 1, 0, _lib.EPI_L2NORM) == 1 and sup(128, 256, 1, 0, _lib.EPI_NONE) == 1
    assert sup(8, 256, 0, 1, _lib.EPI_L2NORM) == 0          # first layer: no blocked L2-norm form
    assert sup(64, 64, 1, 1, _lib.EPI_RELU) == 0 and sup(256, 256, 1, 1, _lib.EPI_RELU_DROPOUT) == 0
    assert sup(7, 100, 0, 0, _lib.EPI_NONE) == 1            # nothing blocked: the plain entry point's job
    assert lib.gsage_project_fwd_blocked(None, None, None, None, None, 10, 64, 64, 1, None, None, None,
                                         1, 1, None) == -2
    # feeder kernel and the no-alloc loss
    assert lib.gsage_alias_sample(None, None, 10, 5, 0, 0, None, None) == -1
    assert lib.gsage_skipgram_noalloc_max_pairs() == 1 << 24
    assert lib.gsage_skipgram_fwd_noalloc(None, 4, 64, None, None, None, (1 << 24) + 1, 5, 5.0, 8.0, 40.0,
                                          None, None, None) == -5
    assert b"no-alloc" in lib.gsage_error_string(-5)


def test_recompute_gather_and_range_entry_points_validate_before_touching_the_device():
    from graphsage_amd import _lib
    lib = _lib.load()
    # which shapes the recomputed aggregation takes
    sup = lib.gsage_segment_gather_hidden1_supported
    assert sup(64, _lib.EPI_RELU) == 1 and sup(128, _lib.EPI_RELU) == 1 and sup(256, _lib.EPI_RELU) == 1
    assert sup(64, _lib.EPI_RELU_DROPOUT) == 1 and sup(128, _lib.EPI_RELU_DROPOUT) == 1 and sup(96, _lib.EPI_RELU_DROPOUT) == 0
    assert sup(96, _lib.EPI_RELU) == 0 and sup(64, _lib.EPI_NONE) == 0 and sup(64, _lib.EPI_L2NORM) == 0
    # sizes: the 64-byte rows + weight planes; the group index (<= N + E/8 slots of 8 bytes + offsets)
    n, e = 10_000_000, 200_000_000
    assert n * 64 <= lib.gsage_segment_gather_hidden1_workspace_bytes(n, e, 64, _lib.EPI_RELU_DROPOUT) <= n * 64 * 1.2
    assert lib.gsage_segment_gather_hidden1_workspace_bytes(-1, e, 64, _lib.EPI_RELU) == 0
    # training on more than 64 columns: 128-byte rows (the masks of the further column groups ride along)
    assert lib.gsage_segment_gather_hidden1_workspace_bytes(n, e, 128, _lib.EPI_RELU_DROPOUT) >= n * 128
    g = lib.gsage_hidden1_groups_bytes(n, e)
    assert (n + e // 8) * 8 <= g <= 2 * (n + e // 8) * 8 + e // 64
    assert lib.gsage_hidden1_groups_bytes(-1, 0) == 0
    # null pointers, unsupported shapes, rows beyond the 32-bit row offsets: refused, nothing launched
    assert lib.gsage_hidden1_groups_build(None, 10, 10, None, 0, None) == -1
    args = [None] * 8
    assert lib.gsage_segment_gather_hidden1(None, 0, *args[:7], 10, 64, _lib.EPI_RELU, 0.0, None, None,
                                            None, 0, None, 0, None) == -1
    # gsage_project_fwd_range: eval epilogues only, in and out words must differ
    assert lib.gsage_project_fwd_range_supported(8, 256, _lib.EPI_RELU) == 1
    assert lib.gsage_project_fwd_range_supported(64, 64, _lib.EPI_RELU) == 1      # its fp16 two-part kernel
    assert lib.gsage_project_fwd_range_supported(64, 128, _lib.EPI_L2NORM) == 1 and \
        lib.gsage_project_fwd_range_supported(96, 64, _lib.EPI_RELU) == 0
    assert lib.gsage_project_fwd_range(None, None, None, None, None, 10, 256, 256, _lib.EPI_RELU_DROPOUT,
                                       None, None, None, 0, 0, None, None, None) == -2
