"""NumPy restatement of the reference's brute-force hot path and its conventions.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference lines it
follows (paths relative to the reference checkout).  Pinned against tests/golden/*.npz, which were
produced by importing the reference's own NumPy implementation (tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

FLT_MAX = np.float32(np.finfo(np.float32).max)


# --------------------------------------------------------------------------------------------
# helpers (src/algorithms/modular.py:109-118)
# --------------------------------------------------------------------------------------------
def safe_normalize(matrix: np.ndarray) -> np.ndarray:
    """Row-normalise; zero-norm rows become zero rows (modular.py:109-111)."""
    norms = np.linalg.norm(matrix, axis=1, keepdims=True)
    return np.divide(matrix, norms, out=np.zeros_like(matrix), where=norms > 0)


def ensure_float32(vectors: np.ndarray) -> np.ndarray:
    """float32 + C-contiguous without needless copies (modular.py:114-118)."""
    if vectors.dtype == np.float32 and vectors.flags["C_CONTIGUOUS"]:
        return vectors
    return np.ascontiguousarray(vectors, dtype=np.float32)


# --------------------------------------------------------------------------------------------
# LinearSearcher (src/algorithms/modular.py:312-390) -- the literal algorithm, small inputs only
# --------------------------------------------------------------------------------------------
def linear_searcher_batch(X: np.ndarray, Q: np.ndarray, k: int, metric: str) -> Tuple[np.ndarray, np.ndarray]:
    """Literal restatement: broadcasted difference for L2, matmul for cosine/ip, argpartition+argsort.

    Returns (distances float32 (Q,k), indices int64 (Q,k)):
      l2 -> Euclidean (sqrt) ascending; cosine/ip -> negated score ascending; k>N -> +inf / -1.
    """
    X = ensure_float32(X)
    Q = np.asarray(Q)
    if Q.ndim == 1:
        Q = Q.reshape(1, -1)
    Q = Q.astype(np.float32, copy=True)
    if metric == "l2":
        diffs = X[None, :, :] - Q[:, None, :]
        vals = np.sum(diffs ** 2, axis=2)
    elif metric in ("cosine", "ip"):
        if metric == "cosine":
            vals = -(safe_normalize(Q) @ safe_normalize(X).T)
        else:
            vals = -(Q @ X.T)
    else:
        raise ValueError(f"Unsupported metric '{metric}' for LinearSearcher")
    if vals.shape[1] == 0:
        raise RuntimeError("LinearSearcher cannot operate on empty index")
    limit = min(k, vals.shape[1])
    kth = max(limit - 1, 0)
    part = np.argpartition(vals, kth=kth, axis=1)[:, :limit]
    rows = np.arange(Q.shape[0])[:, None]
    pv = vals[rows, part]
    order = np.argsort(pv, axis=1)
    idx = part[rows, order]
    dist = pv[rows, order]
    if metric == "l2":
        dist = np.sqrt(dist)
    if limit < k:
        dist = np.pad(dist, ((0, 0), (0, k - limit)), constant_values=np.inf)
        idx = np.pad(idx, ((0, 0), (0, k - limit)), constant_values=-1)
    return dist.astype(np.float32), idx.astype(np.int64)


# --------------------------------------------------------------------------------------------
# convention matrix: "flat" (ExactSearch / faiss.IndexFlat) <-> LinearSearcher
# --------------------------------------------------------------------------------------------
def flat_to_linear(dist_flat: np.ndarray, ids: np.ndarray, metric: str) -> Tuple[np.ndarray, np.ndarray]:
    """Map flat-convention output (squared L2 asc / raw IP desc, FLT_MAX padding) to the
    LinearSearcher convention (sqrt L2 / negated score, +inf padding) -- modular.py:355-360, 381-385."""
    d = np.array(dist_flat, dtype=np.float32, copy=True)
    pad = ids < 0
    if metric == "l2":
        d = np.sqrt(np.where(pad, np.float32(0), d)).astype(np.float32)
    else:
        d = (-d).astype(np.float32)
    d[pad] = np.inf
    return d, ids.astype(np.int64)


def exact_search_metric(metric: str) -> str:
    """ExactSearch maps 'l2' -> METRIC_L2 and ANYTHING else (incl. 'cosine') to raw inner product
    with no normalisation (exact_search.py:23)."""
    return "l2" if metric == "l2" else "ip"


# --------------------------------------------------------------------------------------------
# metrics (src/benchmark/metrics.py:4-34)
# --------------------------------------------------------------------------------------------
def recall_at_k(ground_truth: np.ndarray, predicted: np.ndarray, k: int) -> float:
    if k > predicted.shape[1]:
        k = predicted.shape[1]
    total = 0.0
    for g_row, p_row in zip(ground_truth, predicted):
        g = set(g_row[:k].tolist()) if ground_truth.shape[1] >= k else set(g_row.tolist())
        p = set(p_row[:k].tolist())
        total += (len(g & p) / len(g)) if g else 0.0
    return total / ground_truth.shape[0]


# --------------------------------------------------------------------------------------------
# synthetic data exactly as the reference generates it (src/benchmark/dataset.py:473-504)
# --------------------------------------------------------------------------------------------
def random_dataset(dimensions: int = 128, train_size: int = 10_000, test_size: int = 1_000, seed: int = 42):
    """np.random.seed(seed); train = randn(train,dim).f32; test = randn(test,dim).f32 (same stream)."""
    state = np.random.get_state()
    try:
        np.random.seed(seed)
        X = np.random.randn(train_size, dimensions).astype(np.float32)
        Q = np.random.randn(test_size, dimensions).astype(np.float32)
    finally:
        np.random.set_state(state)
    return X, Q


def ground_truth_l2(X: np.ndarray, Q: np.ndarray, k: int) -> np.ndarray:
    """argsort(norm(X - q))[:k] per query, int32 (dataset.py:497-504)."""
    gt = np.zeros((Q.shape[0], k), dtype=np.int32)
    for i in range(Q.shape[0]):
        gt[i] = np.argsort(np.linalg.norm(X - Q[i:i + 1], axis=1))[:k]
    return gt
