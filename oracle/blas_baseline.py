"""CPU baselines of SURVEY 8(d), timed by bench.py's `cpu_baseline` leg (TEST INFRASTRUCTURE ONLY).

  gemm_expansion_knn   (ii) the realistic CPU competitor: what faiss.IndexFlat.search does on behalf of
                       ExactSearch.batch_search (exact_search.py:62-78) -- float32 GEMM expansion
                       ||x||^2 - 2 q.x (+ ||q||^2) or q.x through the threaded BLAS NumPy links (OpenBLAS), query
                       block x corpus block, per-block `argpartition` + running top-k merge (FAISS: sgemm blocks of
                       4096 x 1024 + heap updates, OpenMP over queries).  The select is spread over a thread pool
                       (NumPy's partition releases the GIL), the GEMM uses the BLAS's own threads.
  linear_searcher_qps  (i) the reference's NumPy `LinearSearcher` literally (oracle/ref_semantics.py restating
                       modular.py:341-360): direct-difference L2 with Q_b x N x D temporaries, which is why the
                       query batch has to be tiny at 1M rows.
Neither is used as a checker of values (that is knn_oracle.c MODE_CANON); their ids are compared with the GPU's
only as a recall sanity check.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Tuple

import numpy as np


def blas_info() -> dict:
    try:
        from threadpoolctl import threadpool_info

        for lib in threadpool_info():
            if lib.get("user_api") == "blas":
                return {"blas": lib.get("internal_api"), "blas_threads": lib.get("num_threads"),
                        "blas_version": lib.get("version")}
    except Exception:  # noqa: BLE001
        pass
    return {"blas": "unknown", "blas_threads": None}


def _topk_rows(scores: np.ndarray, k: int, lo: int, hi: int, out_v: np.ndarray, out_i: np.ndarray) -> None:
    s = scores[lo:hi]
    part = np.argpartition(s, k - 1, axis=1)[:, :k]
    out_i[lo:hi] = part
    out_v[lo:hi] = np.take_along_axis(s, part, axis=1)


def gemm_expansion_knn(X: np.ndarray, Q: np.ndarray, k: int, metric: str, qblock: int = 1024, nblock: int = 65536,
                       select_threads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Flat conventions: squared L2 ascending / raw inner product descending, int64 ids."""
    n, nq = X.shape[0], Q.shape[0]
    threads = select_threads or (os.cpu_count() or 1)
    xn = np.einsum("ij,ij->i", X, X) if metric == "l2" else None
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    with ThreadPoolExecutor(threads) as pool:
        for q0 in range(0, nq, qblock):
            q = Q[q0:q0 + qblock]
            m = q.shape[0]
            best_v = np.full((m, 0), 0, np.float32)
            best_i = np.full((m, 0), 0, np.int64)
            for n0 in range(0, n, nblock):
                xb = X[n0:n0 + nblock]
                s = q @ xb.T                                       # threaded sgemm
                if metric == "l2":
                    s *= -2.0
                    s += xn[None, n0:n0 + nblock]
                else:
                    np.negative(s, out=s)
                kk = min(k, s.shape[1])
                v = np.empty((m, kk), np.float32)
                i = np.empty((m, kk), np.int64)
                step = max(1, -(-m // threads))
                list(pool.map(lambda lo: _topk_rows(s, kk, lo, min(m, lo + step), v, i), range(0, m, step)))
                best_v = np.concatenate([best_v, v], axis=1)
                best_i = np.concatenate([best_i, i + n0], axis=1)
                if best_v.shape[1] > 4 * k:                          # running merge
                    keep = np.argpartition(best_v, k - 1, axis=1)[:, :k]
                    best_v = np.take_along_axis(best_v, keep, axis=1)
                    best_i = np.take_along_axis(best_i, keep, axis=1)
            kk = min(k, best_v.shape[1])
            order = np.argsort(best_v, axis=1, kind="stable")[:, :kk]
            bv = np.take_along_axis(best_v, order, axis=1)
            bi = np.take_along_axis(best_i, order, axis=1)
            if metric == "l2":
                bv = bv + np.einsum("ij,ij->i", q, q)[:, None]
            else:
                bv = -bv
            D[q0:q0 + m, :kk] = bv
            I[q0:q0 + m, :kk] = bi
            if kk < k:
                D[q0:q0 + m, kk:] = np.finfo(np.float32).max if metric == "l2" else -np.finfo(np.float32).max
                I[q0:q0 + m, kk:] = -1
    return D, I


def time_gemm_expansion(X, Q, k, metric, budget_s: float = 12.0):
    """(result dict, ids of the timed sample): warm the BLAS threads on one block, then time a bounded sample."""
    info = blas_info()
    probe = min(len(Q), 1024)
    t0 = time.perf_counter()
    gemm_expansion_knn(X, Q[:probe], k, metric)
    dt = time.perf_counter() - t0
    sample = int(min(len(Q), max(probe, (budget_s / max(dt, 1e-6)) * probe)))
    sample = max(1024, sample // 1024 * 1024) if len(Q) >= 1024 else len(Q)
    sample = min(sample, len(Q))
    t0 = time.perf_counter()
    _, ids = gemm_expansion_knn(X, Q[:sample], k, metric)
    dt = time.perf_counter() - t0
    return {"value": round(sample / dt, 2), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
            "impl": "oracle/blas_baseline.py gemm_expansion_knn: float32 -2QX^T+norms via NumPy's threaded "
                    f"{info.get('blas')} ({info.get('blas_threads')} BLAS threads), 1024 x 65536 blocks, "
                    f"argpartition on {os.cpu_count()} threads",
            "sample": f"first {sample} of {len(Q)} queries against all {len(X)} rows, {dt:.1f} s", **info}, ids


def time_linear_searcher(X, Q, k, metric, qbatch: int = 4, budget_s: float = 6.0):
    """The reference's NumPy LinearSearcher restated literally; query_batch_size such that the Q_b x N x D float32
    temporaries stay a few GB."""
    from . import ref_semantics as rs

    done, t0 = 0, time.perf_counter()
    while done < len(Q) and (done == 0 or time.perf_counter() - t0 < budget_s):
        rs.linear_searcher_batch(X, Q[done:done + qbatch], k, metric)
        done += qbatch
    dt = time.perf_counter() - t0
    done = min(done, len(Q))
    return {"value": round(done / dt, 3), "unit": "queries/s", "cores": 1, "kind": "port",
            "impl": "oracle/ref_semantics.py linear_searcher_batch (NumPy broadcast difference, modular.py:341-360), "
                    f"query_batch_size {qbatch}",
            "sample": f"first {done} of {len(Q)} queries against all {len(X)} rows, {dt:.1f} s"}
