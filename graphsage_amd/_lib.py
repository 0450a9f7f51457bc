"""ctypes binding of libgsage.so (the C ABI declared in include/gsage.h).

The product path has no CPU or library fallback: if the HIP library is missing the
import of :mod:`graphsage_amd` still succeeds (so CPU-only tooling can inspect the
package), but the first call that needs a kernel raises ``GsageLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgsage.so")

EPI_NONE, EPI_RELU, EPI_RELU_DROPOUT, EPI_L2NORM = 0, 1, 2, 3
FLAG_INDEX_OUT_OF_RANGE = 1


class GsageLibraryError(RuntimeError):
    pass


_p = C.c_void_p
_i64 = C.c_int64
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/gsage.h one to one
SIGNATURES = {
    "gsage_version": (C.c_int, []),
    "gsage_error_string": (C.c_char_p, [C.c_int]),
    "gsage_csr_workspace_bytes": (_sz, [_i64, _i64]),
    "gsage_long_row_threshold": (C.c_int, []),
    "gsage_csr_build": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p,
                                  _sz, _p]),
    "gsage_segment_gather": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "gsage_segment_gather_skip": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "gsage_segment_gather_live": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "gsage_segment_gather_hidden1_supported": (C.c_int, [_i64, C.c_int]),
    "gsage_segment_gather_hidden1_workspace_bytes": (_sz, [_i64, _i64, _i64, C.c_int]),
    "gsage_hidden1_groups_bytes": (_sz, [_i64, _i64]),
    "gsage_hidden1_groups_build": (C.c_int, [_p, _i64, _i64, _p, _sz, _p]),
    "gsage_segment_gather_hidden1": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, C.c_int,
                                               C.c_float, _p, _p, _p, C.c_int, _p, _sz, _p]),
    "gsage_hidden1_u_row_bytes": (_sz, [_i64, C.c_int]),
    "gsage_hidden1_u_rows": (_i64, [_i64]),
    "gsage_segment_gather_hidden1_u_workspace_bytes": (_sz, [_i64, _i64]),
    "gsage_segment_gather_hidden1_u": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, C.c_int,
                                                 C.c_float, _p, C.c_int, _p, _sz, _p]),
    "gsage_project_fwd_u": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int, _p,
                                      C.c_uint64, C.c_float, _p, _p, _p, C.c_int, _p, _p,
                                      C.POINTER(C.c_int), _p, _p, _p]),
    "gsage_project_bwd_rows_supported": (C.c_int, [_i64, _i64]),
    "gsage_project_bwd_rows": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int,
                                         C.c_float, _p, _p, _p, _p, _p, _p, _sz, _p, _p, _i64, _p]),
    "gsage_project_bwd_rows_range": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int,
                                               C.c_float, _p, _p, _p, _p, _p, _p, _sz, _p, _p, _i64, _p, _p]),
    "gsage_hidden1_wgrad_supported": (C.c_int, [_i64, C.c_int]),
    "gsage_hidden1_wgrad_workspace_bytes": (_sz, [_i64, _i64]),
    "gsage_hidden1_wgrad": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, C.c_int,
                                      C.c_float, _p, _p, _p, _p, _sz, _p]),
    "gsage_touched_rows_workspace_bytes": (_sz, [_i64]),
    "gsage_touched_rows": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _i64, _p, _p, _sz, _p]),
    "gsage_zero_rows": (C.c_int, [_p, _p, _i64, _i64, _p, _p]),
    "gsage_touched_rows_ex": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    "gsage_skipgram_bwd_rows_supported": (C.c_int, [_i64]),
    "gsage_skipgram_bwd_rows_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gsage_skipgram_bwd_rows": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p,
                                          _p, _p, _p, _p, _i64, _p, _sz, _p]),
    "gsage_project_workspace_bytes": (_sz, [_i64, _i64]),
    "gsage_project_fwd": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int, _p,
                                    C.c_uint64, C.c_float, _p, _p, _p, C.c_int, _p]),
    "gsage_project_fwd_blocked_supported": (C.c_int, [_i64, _i64, C.c_int, C.c_int, C.c_int]),
    "gsage_project_fwd_blocked": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int, _p, _p,
                                            _p, C.c_int, C.c_int, _p]),
    "gsage_project_fwd_range_supported": (C.c_int, [_i64, _i64, C.c_int]),
    "gsage_project_fwd_range": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int, _p, _p,
                                          _p, C.c_int, C.c_int, _p, _p, _p]),
    "gsage_project_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gsage_project_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int,
                                    C.c_float, _p, _p, _p, _p, _p, _p, _p, _sz, C.c_int, _p]),
    "gsage_relu_sign_bytes": (_sz, [_i64, _i64]),
    "gsage_project_fwd_ex": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int, _p,
                                       C.c_uint64, C.c_float, _p, _p, _p, C.c_int, _p, _p]),
    "gsage_project_bwd_ex": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_int,
                                       C.c_float, _p, _p, _p, _p, _p, _p, _p, _sz, C.c_int, _p, _p,
                                       _p]),
    "gsage_skipgram_workspace_bytes": (_sz, [_i64]),
    "gsage_skipgram_fwd": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _i64, _i64, C.c_float,
                                     C.c_double, C.c_double, _p, _p, _p, _sz, _p]),
    "gsage_skipgram_noalloc_max_pairs": (_i64, []),
    "gsage_skipgram_fwd_noalloc": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _i64, _i64, C.c_float,
                                             C.c_double, C.c_double, _p, _p, _p]),
    "gsage_skipgram_bwd": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "gsage_alias_build_host": (C.c_int, [_p, _i64, _p, _p]),
    "gsage_alias_sample": (C.c_int, [_p, _p, _i64, _i64, C.c_uint64, C.c_uint64, _p, _p]),
    "gsage_random_walks": (C.c_int, [_p, _p, _i64, _i64, _i64, C.c_uint64, _p, _p, _p]),
    "gsage_walk_pair_counts": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "gsage_walk_pairs_gather": (C.c_int, [_p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _p]),
    "gsage_segment_gather_ew": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "gsage_skipgram_bwd_det_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gsage_skipgram_bwd_det": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p, _p,
                                         _sz, _p]),
}

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load libgsage.so once; raise loudly when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise GsageLibraryError(
            f"{_LIB_PATH} not found: build it with `make -C graphsage_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "graphsage_amd has no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise GsageLibraryError(f"libgsage.so lacks symbol {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().gsage_error_string(status).decode()
        exc = ValueError if status in (-1, -2, -5) else RuntimeError
        raise exc(f"{what}: {msg} (gsage status {status})")


def ptr(t) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_handle() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
