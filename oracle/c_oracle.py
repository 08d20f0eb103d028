"""ctypes front-end of oracle/knn_oracle.c (TEST INFRASTRUCTURE ONLY).

`knn(X, Q, k, metric, mode=...)` returns flat-convention results (squared L2 ascending / raw inner
product descending, int64 ids, -1 / +-FLT_MAX padding), the convention of the reference's
ExactSearch (exact_search.py:62-78 -> faiss.IndexFlat.search).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liboracle.so"
_lib: Optional[ctypes.CDLL] = None

MODE_CANON, MODE_NUMPY32, MODE_GEMM32 = 0, 1, 2
_METRIC = {"l2": 0, "ip": 1}


def build(force: bool = False) -> Path:
    """Compile the C oracle with gcc (seconds).  Building the checker is not using it."""
    srcs = sorted(_HERE.glob("*.c"))
    stale = (not _LIB_PATH.exists()) or any(s.stat().st_mtime > _LIB_PATH.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        L = ctypes.CDLL(str(_LIB_PATH))
        f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.oracle_knn.restype = ctypes.c_int
        L.oracle_knn.argtypes = [f32, ctypes.c_int64, ctypes.c_int, f32, ctypes.c_int64, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int64, f32, ctypes.c_void_p, i64,
                                 ctypes.c_int]
        L.oracle_knn_gemm32.restype = ctypes.c_int
        L.oracle_knn_gemm32.argtypes = [f32, ctypes.c_int64, ctypes.c_int, f32, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int64, f32, i64, ctypes.c_int]
        L.oracle_pair_keys.restype = ctypes.c_int
        L.oracle_pair_keys.argtypes = [f32, ctypes.c_int, f32, ctypes.c_int64, ctypes.c_int, ctypes.c_int, i64, f64]
        L.oracle_merge_partials.restype = ctypes.c_int
        L.oracle_merge_partials.argtypes = [f64, i64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                            f32, i64]
        L.oracle_num_threads.restype = ctypes.c_int
        i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        L.oracle_ivf_assign.restype = ctypes.c_int
        L.oracle_ivf_assign.argtypes = [f32, ctypes.c_int, ctypes.c_int, f32, ctypes.c_int64, ctypes.c_int, i32,
                                        ctypes.c_int]
        L.oracle_ivf_search.restype = ctypes.c_int
        L.oracle_ivf_search.argtypes = [f32, ctypes.c_int64, ctypes.c_int, f32, ctypes.c_int, i32, f32, ctypes.c_int64,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, f32, i64, ctypes.c_int]
        L.oracle_kmeans_objective.restype = ctypes.c_double
        L.oracle_kmeans_objective.argtypes = [f32, ctypes.c_int, ctypes.c_int, f32, ctypes.c_int64, ctypes.c_int]
        _lib = L
    return _lib


def _prep(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def knn(X: np.ndarray, Q: np.ndarray, k: int, metric: str = "l2", mode: int = MODE_CANON, id_base: int = 0,
        threads: int = 0, return_keys: bool = False):
    """Exhaustive exact k-NN.  mode: MODE_CANON (float64, the product's arithmetic),
    MODE_NUMPY32 (bit-faithful to NumPy's float32 L2 branch), MODE_GEMM32 (FAISS-style expansion)."""
    X, Q = _prep(X), _prep(Q)
    if Q.ndim == 1:
        Q = Q.reshape(1, -1)
    n, D = X.shape if X.ndim == 2 else (0, Q.shape[1])
    nq = Q.shape[0]
    dist = np.empty((nq, k), np.float32)
    ids = np.empty((nq, k), np.int64)
    m = _METRIC[metric]
    if mode == MODE_GEMM32:
        rc = lib().oracle_knn_gemm32(X, n, D, Q, nq, k, m, id_base, dist, ids, threads)
        keys = None
    else:
        keys = np.empty((nq, k), np.float64) if return_keys else None
        kp = keys.ctypes.data_as(ctypes.c_void_p) if keys is not None else None
        rc = lib().oracle_knn(X, n, D, Q, nq, k, m, mode, id_base, dist, kp, ids, threads)
    if rc != 0:
        raise RuntimeError(f"oracle_knn failed with code {rc}")
    return (dist, ids, keys) if return_keys else (dist, ids)


def pair_keys(X: np.ndarray, Q: np.ndarray, ids: np.ndarray, metric: str = "l2") -> np.ndarray:
    """Canonical float64 sort keys (L2: squared distance; IP: -score) of the given (query,row) pairs."""
    X, Q = _prep(X), _prep(Q)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    out = np.empty(ids.shape, np.float64)
    lib().oracle_pair_keys(X, X.shape[1], Q, Q.shape[0], ids.shape[1], _METRIC[metric], ids, out)
    return out


def merge_partials(keys: np.ndarray, ids: np.ndarray, metric: str = "l2") -> Tuple[np.ndarray, np.ndarray]:
    """Merge (nparts, nq, k) per-shard partial lists into the global (nq, k) result."""
    keys = np.ascontiguousarray(keys, np.float64)
    ids = np.ascontiguousarray(ids, np.int64)
    nparts, nq, k = keys.shape
    dist = np.empty((nq, k), np.float32)
    out = np.empty((nq, k), np.int64)
    rc = lib().oracle_merge_partials(keys, ids, nparts, nq, k, _METRIC[metric], dist, out)
    if rc != 0:
        raise RuntimeError(f"oracle_merge_partials failed with code {rc}")
    return dist, out


def num_threads() -> int:
    return int(lib().oracle_num_threads())


# ---- IVF-Flat (oracle/ivf_oracle.c) ------------------------------------------------------------------
def ivf_assign(C: np.ndarray, X: np.ndarray, metric: str = "l2", threads: int = 0) -> np.ndarray:
    """List (nearest centroid under the index metric, canonical arithmetic) of every row."""
    C, X = _prep(C), _prep(X)
    out = np.empty((X.shape[0],), np.int32)
    rc = lib().oracle_ivf_assign(C, C.shape[0], C.shape[1], X, X.shape[0], _METRIC[metric], out, threads)
    if rc != 0:
        raise RuntimeError(f"oracle_ivf_assign failed with code {rc}")
    return out


def ivf_search(X, C, list_of_row, Q, k: int, nprobe: int, metric: str = "l2", id_base: int = 0, threads: int = 0):
    """IVF-Flat search = brute force restricted to the nprobe nearest lists (flat conventions)."""
    X, C, Q = _prep(X), _prep(C), _prep(Q)
    lor = np.ascontiguousarray(list_of_row, np.int32)
    dist = np.empty((Q.shape[0], k), np.float32)
    ids = np.empty((Q.shape[0], k), np.int64)
    rc = lib().oracle_ivf_search(X, X.shape[0], X.shape[1], C, C.shape[0], lor, Q, Q.shape[0], k, nprobe,
                                 _METRIC[metric], id_base, dist, ids, threads)
    if rc != 0:
        raise RuntimeError(f"oracle_ivf_search failed with code {rc}")
    return dist, ids


def kmeans_objective(C, X, threads: int = 0) -> float:
    """Mean squared distance of every row to its nearest centroid."""
    C, X = _prep(C), _prep(X)
    return float(lib().oracle_kmeans_objective(C, C.shape[0], C.shape[1], X, X.shape[0], threads))
