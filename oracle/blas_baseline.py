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


def usable_cpus() -> int:
    """CPUs this process may actually use: the cgroup quota when there is one (a GPU box gives one GPU's share of a
    256-thread host), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, n)


def _first_block(scores, k, lo, hi, out_v, out_i):
    s = scores[lo:hi]
    part = np.argpartition(s, k - 1, axis=1)[:, :k]
    out_i[lo:hi] = part
    out_v[lo:hi] = np.take_along_axis(s, part, axis=1)


def _filter_block(scores, thr, lo, hi, chunk=1024):
    """(row, col, value) of the scores below their row's current k-th best -- the compare a heap top does.  Two levels:
    minima of 1024-column chunks first (one vectorised pass), then only the chunks that can hold a passer."""
    s, t = scores[lo:hi], thr[lo:hi]
    mm, nb = s.shape
    nc = nb // chunk
    rows, cols, vals = [], [], []
    if nc:
        s3 = s[:, :nc * chunk].reshape(mm, nc, chunk)
        rr, cc = np.nonzero(s3.min(axis=2) < t[:, None])
        if rr.size:
            sub = s3[rr, cc]                                   # (pairs, chunk)
            pr, pc = np.nonzero(sub < t[rr][:, None])
            rows.append(rr[pr] + lo)
            cols.append(cc[pr] * chunk + pc)
            vals.append(sub[pr, pc])
    if nc * chunk < nb:
        tail = s[:, nc * chunk:]
        r, c = np.nonzero(tail < t[:, None])
        rows.append(r + lo)
        cols.append(c + nc * chunk)
        vals.append(tail[r, c])
    if not rows:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32)
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def gemm_expansion_knn(X: np.ndarray, Q: np.ndarray, k: int, metric: str, qblock: int = 1024, nblock: int = 32768,
                       select_threads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Flat conventions: squared L2 ascending / raw inner product descending, int64 ids.

    One sgemm per (query block, corpus block) on augmented operands -- L2: [-2q, 1] . [x, ||x||^2], IP: [-q] . [x] --
    so the GEMM output IS the order key; the first corpus block seeds every query's top-k with argpartition, every
    later block only keeps scores below the query's current k-th best (FAISS's heap-top compare), merged per block."""
    n, nq = X.shape[0], Q.shape[0]
    threads = select_threads or usable_cpus()
    if metric == "l2":
        Xa = np.empty((n, X.shape[1] + 1), np.float32)
        Xa[:, :-1] = X
        Xa[:, -1] = np.einsum("ij,ij->i", X, X)
    else:
        Xa = X
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    with ThreadPoolExecutor(threads) as pool:
        for q0 in range(0, nq, qblock):
            q = Q[q0:q0 + qblock]
            m = q.shape[0]
            if metric == "l2":
                qa = np.empty((m, q.shape[1] + 1), np.float32)
                qa[:, :-1] = -2.0 * q
                qa[:, -1] = 1.0
            else:
                qa = -q
            step = max(1, -(-m // threads))
            chunks = list(range(0, m, step))
            kk = min(k, n)
            best_v = np.empty((m, kk), np.float32)
            best_i = np.empty((m, kk), np.int64)
            sbuf = np.empty((m, min(nblock, n)), np.float32)       # reused: a fresh 128 MB result per block would
            for n0 in range(0, n, nblock):                          # spend more time in page faults than in sgemm
                xb = Xa[n0:n0 + nblock]
                s = sbuf[:, :xb.shape[0]]
                if s.flags["C_CONTIGUOUS"]:
                    np.matmul(qa, xb.T, out=s)                      # threaded sgemm
                else:
                    s = qa @ xb.T
                if n0 == 0 and s.shape[1] >= kk:
                    list(pool.map(lambda lo: _first_block(s, kk, lo, min(m, lo + step), best_v, best_i), chunks))
                    continue
                thr = best_v.max(axis=1) if n0 else np.full((m,), np.inf, np.float32)
                parts = list(pool.map(lambda lo: _filter_block(s, thr, lo, min(m, lo + step)), chunks))
                r = np.concatenate([p[0] for p in parts])
                if r.size == 0:
                    continue
                c = np.concatenate([p[1] for p in parts]) + n0
                v = np.concatenate([p[2] for p in parts])
                rows = np.concatenate([np.repeat(np.arange(m), best_v.shape[1]), r])
                vals = np.concatenate([best_v.ravel(), v])
                ids = np.concatenate([best_i.ravel(), c])
                order = np.lexsort((vals, rows))                     # by row, then value
                rows, vals, ids = rows[order], vals[order], ids[order]
                first = np.searchsorted(rows, np.arange(m))
                rank = np.arange(rows.size) - first[rows]
                keep = rank < kk
                best_v = vals[keep].reshape(m, kk)
                best_i = ids[keep].reshape(m, kk)
            order = np.argsort(best_v, axis=1, kind="stable")
            bv = np.take_along_axis(best_v, order, axis=1)
            bi = np.take_along_axis(best_i, order, axis=1)
            bv = bv + np.einsum("ij,ij->i", q, q)[:, None] if metric == "l2" else -bv
            D[q0:q0 + m, :kk] = bv
            I[q0:q0 + m, :kk] = bi
            if kk < k:
                D[q0:q0 + m, kk:] = np.finfo(np.float32).max if metric == "l2" else -np.finfo(np.float32).max
                I[q0:q0 + m, kk:] = -1
    return D, I


def time_gemm_expansion(X, Q, k, metric, budget_s: float = 12.0, threads: int = 0):
    """(result dict, ids of the timed sample): warm the BLAS threads on one block, then time a bounded sample.
    `threads` = select threads AND the number the caller has limited the BLAS to (bench.py: threadpool_limits) -- one
    figure, reported as `cores`; 0 = usable_cpus()."""
    info = blas_info()
    cpus = threads or usable_cpus()
    probe = min(len(Q), 1024)
    t0 = time.perf_counter()
    gemm_expansion_knn(X, Q[:probe], k, metric, select_threads=cpus)
    dt = time.perf_counter() - t0
    sample = int(min(len(Q), max(probe, (budget_s / max(dt, 1e-6)) * probe)))
    sample = max(1024, sample // 1024 * 1024) if len(Q) >= 1024 else len(Q)
    sample = min(sample, len(Q))
    t0 = time.perf_counter()
    _, ids = gemm_expansion_knn(X, Q[:sample], k, metric, select_threads=cpus)
    dt = time.perf_counter() - t0
    return {"value": round(sample / dt, 2), "unit": "queries/s", "cores": cpus, "kind": "port",
            "impl": "oracle/blas_baseline.py gemm_expansion_knn: float32 GEMM expansion via NumPy's threaded "
                    f"{info.get('blas')} ({info.get('blas_threads')} BLAS threads, {cpus} CPUs usable of "
                    f"{os.cpu_count()}), 1024 x 32768 blocks, heap-top filter + per-block merge on {cpus} threads",
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
