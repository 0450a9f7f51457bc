"""Host mirror of the in-kernel dropout hash (csrc/project.hip: dropout_keep)."""
import numpy as np


def dropout_keep_mask(seed: int, n_rows: int, n_cols: int, p: float) -> np.ndarray:
    """uint8 [n_rows, n_cols]; 1 where the kernel keeps element (row, col)."""
    seed &= 0xFFFFFFFFFFFFFFFF
    idx = np.arange(n_rows * n_cols, dtype=np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = np.uint32(seed & 0xFFFFFFFF) ^ (lo * np.uint32(0x9E3779B1))
        h = h ^ (np.uint32(seed >> 32) + hi * np.uint32(0x85EBCA77))
        h = h ^ (h >> np.uint32(16))
        h = h * np.uint32(0x85EBCA6B)
        h = h ^ (h >> np.uint32(13))
        h = h * np.uint32(0xC2B2AE35)
        h = h ^ (h >> np.uint32(16))
    u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).astype(np.uint8).reshape(n_rows, n_cols)
