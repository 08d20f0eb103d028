"""On-disk formats either side of the hot path.

  read_fvecs / read_ivecs   TEXMEX .fvecs / .ivecs (SIFT1M): every record is <int32 dim><dim x 4-byte values>.
                            The reference's `_read_fvecs` (src/benchmark/dataset.py:522-547) reads the payload as
                            int32 and VALUE-casts it to float32 (a stored 1.0f loads as 1.0653532e9); here the
                            payload bytes are re-interpreted (`.view(float32)`), which is what the format means.
                            `_read_ivecs` (:549-574) is correct in the reference and behaves the same here.
  open_npy_rows             memory-mapped .npy corpus: FlatIndex.add() uploads straight from the mapping (the
                            pages stream through the HIP staging buffers), so a large corpus is never duplicated
                            in host RAM (the reference's memmap cache: dataset.py:376-471, 1001-1052).
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional

import numpy as np


def _read_vecs(path, dtype, limit: Optional[int]) -> np.ndarray:
    path = Path(path)
    raw = np.memmap(path, dtype=np.int32, mode="r")
    if raw.size == 0:
        return np.zeros((0, 0), dtype)
    dim = int(raw[0])
    if dim <= 0 or raw.size % (dim + 1) != 0:
        raise ValueError(f"{path} is not a valid .{'f' if dtype == np.float32 else 'i'}vecs file (dim field {dim})")
    rec = raw.reshape(-1, dim + 1)
    if limit is not None:
        rec = rec[:limit]
    if not np.all(rec[:, 0] == dim):
        raise ValueError(f"{path}: records with differing dimensions")
    body = np.ascontiguousarray(rec[:, 1:])
    return body.view(np.float32) if dtype == np.float32 else body


def read_fvecs(path, limit: Optional[int] = None) -> np.ndarray:
    return _read_vecs(path, np.float32, limit)


def read_ivecs(path, limit: Optional[int] = None) -> np.ndarray:
    return _read_vecs(path, np.int32, limit)


def write_fvecs(path, x: np.ndarray) -> None:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty((x.shape[0], x.shape[1] + 1), np.int32)
    out[:, 0] = x.shape[1]
    out[:, 1:] = x.view(np.int32)
    out.tofile(path)


def write_ivecs(path, x: np.ndarray) -> None:
    x = np.ascontiguousarray(x, np.int32)
    out = np.empty((x.shape[0], x.shape[1] + 1), np.int32)
    out[:, 0] = x.shape[1]
    out[:, 1:] = x
    out.tofile(path)


def open_npy_rows(path, limit: Optional[int] = None) -> np.ndarray:
    """Read-only memory map of a 2-D float32 .npy file (optionally its first `limit` rows)."""
    arr = np.load(path, mmap_mode="r")
    if arr.ndim != 2:
        raise ValueError(f"{path}: expected a 2-D array, got shape {arr.shape}")
    return arr[:limit] if limit is not None else arr
