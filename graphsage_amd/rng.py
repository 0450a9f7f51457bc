"""Host mirror of the in-kernel dropout hash (csrc/project.hip: dropout_keep)."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def dropout_keep_mask(seed: int, n_rows: int, n_cols: int, p: float) -> np.ndarray:
    """uint8 [n_rows, n_cols]; 1 where the kernel keeps element (row, col)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n_rows * n_cols, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).astype(np.uint8).reshape(n_rows, n_cols)
