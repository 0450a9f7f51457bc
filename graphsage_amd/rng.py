"""Host mirrors of the in-kernel counter hashes: the dropout mask (csrc/project.hip:
dropout_keep) and the alias-method draws (csrc/walks.hip: alias_sample_k)."""
import numpy as np


def dropout_keep_mask(seed: int, n_rows: int, n_cols: int, p: float) -> np.ndarray:
    """uint8 [n_rows, n_cols]; 1 where the kernel keeps element (row, col)."""
    return dropout_keep_mask_range(seed, 0, n_rows * n_cols, p).reshape(n_rows, n_cols)


def dropout_keep_mask_range(seed: int, first: int, count: int, p: float) -> np.ndarray:
    """uint8 [count]: elements first .. first + count of the row-major [N, F] activation."""
    seed &= 0xFFFFFFFFFFFFFFFF
    idx = np.arange(count, dtype=np.uint64) + np.uint64(first)
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
    return (u >= np.float32(p)).astype(np.uint8)


def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h ^ (h >> np.uint32(16))
    h = h * np.uint32(0x85EBCA6B)
    h = h ^ (h >> np.uint32(13))
    h = h * np.uint32(0xC2B2AE35)
    return h ^ (h >> np.uint32(16))


def alias_draws(seed: int, first_draw: int, count: int, n: int):
    """(bin int64 [count], u float32 [count]) exactly as alias_sample_k draws them."""
    seed &= 0xFFFFFFFFFFFFFFFF
    ctr = np.arange(count, dtype=np.uint64) + np.uint64(first_draw)
    lo = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (ctr >> np.uint64(32)).astype(np.uint32)
    s_lo, s_hi = np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32)
    with np.errstate(over="ignore"):
        a = _fmix32((s_lo ^ (lo * np.uint32(0x9E3779B1))) ^ (s_hi + hi * np.uint32(0x85EBCA77)))
        b = _fmix32(a ^ np.uint32(0x68E31DA4) ^ (lo * np.uint32(0xB5297A4D)))
        c = _fmix32((b + np.uint32(0x1B56C4E9) + s_hi) ^ hi)
        nn = np.uint64(n)
        bins = (a.astype(np.uint64) * nn + ((c.astype(np.uint64) * nn) >> np.uint64(32))) >> np.uint64(32)
    u = (b >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return bins.astype(np.int64), u


def alias_sample(prob_table: np.ndarray, alias: np.ndarray, seed: int, first_draw: int,
                 count: int) -> np.ndarray:
    """The negatives gsage_alias_sample writes for the same table, seed and counters."""
    bins, u = alias_draws(seed, first_draw, count, prob_table.shape[0])
    return np.where(u >= prob_table[bins], alias[bins], bins).astype(np.int64)
