"""Synthetic corpora of the BASELINE.json shapes (no dataset files exist offline).

  random_reference   the reference's own generator (src/benchmark/dataset.py:487-495): one global
                     NumPy stream seeded once; train then test.
  sift_like          SIFT1M-shaped: non-negative integer-valued float32 in [0, 218], gamma-skewed,
                     rows rescaled to ||x|| ~ 512 and rounded (real SIFT descriptors are quantised
                     to integers and normalised to 512).
  gaussian           standard normal float32 (exercises non-representable fp32 values)
  glove_like         0.5 * standard normal, D = 50 (GloVe-50 shape)
"""
from __future__ import annotations

import numpy as np


def random_reference(dimensions=128, train_size=10_000, test_size=1_000, seed=42):
    state = np.random.get_state()
    try:
        np.random.seed(seed)
        train = np.random.randn(train_size, dimensions).astype(np.float32)
        test = np.random.randn(test_size, dimensions).astype(np.float32)
    finally:
        np.random.set_state(state)
    return train, test


def _sift_rows(rng: np.random.Generator, n: int, dim: int, chunk: int = 262144) -> np.ndarray:
    out = np.empty((n, dim), np.float32)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        v = np.clip(np.rint(rng.gamma(0.6, 40.0, size=(m, dim))), 0, 218).astype(np.float32)
        norm = np.linalg.norm(v, axis=1, keepdims=True)
        norm[norm == 0] = 1.0
        out[s:s + m] = np.clip(np.rint(v * (512.0 / norm)), 0, 218)
    return out


def sift_like(n=1_000_000, nq=10_000, dim=128, seed=1234):
    return _sift_rows(np.random.default_rng(seed), n, dim), _sift_rows(np.random.default_rng(seed + 1), nq, dim)


def gaussian(n=1_000_000, nq=10_000, dim=128, seed=1234, scale=1.0):
    x = np.random.default_rng(seed).standard_normal((n, dim), dtype=np.float32)
    q = np.random.default_rng(seed + 1).standard_normal((nq, dim), dtype=np.float32)
    if scale != 1.0:
        x *= np.float32(scale)
        q *= np.float32(scale)
    return x, q


def glove_like(n=1_200_000, nq=10_000, dim=50, seed=50):
    return gaussian(n, nq, dim, seed, scale=0.5)
