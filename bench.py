#!/usr/bin/env python3
"""bench.py -- QPS of the exact / IVF-Flat k-NN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--nprobe P] [--scaling weak|strong]

A "step" is one pass of the hot path over one 10 000-query batch already resident in HBM: libvdbhip's device
pipeline (query prep -> MFMA scan + bin select -> exact refine), and for N > 1 the RCCL all-gather of the per-shard
partial top-k plus the merge kernel.

N = 1   headline = BASELINE.json configs[1]: SIFT1M-shaped corpus (1 000 000 x 128, byte-valued float32), 10 000
        queries, k = 10, L2 (real SIFT1M files under $VDBHIP_DATA are used when present).  `value` is the
        device-resident rate the bench contract asks for; `value_host_io` beside it is what the reference harness
        measures (wall clock of `batch_search` with NumPy queries in and NumPy (D, I) out,
        experiment_runner.py:428-439, 464).  The SAME line carries every other single-GPU BASELINE config as
        `also.*`, each with its own `roofline`, recall and `cpu_baseline`:
          also.gaussian1m   configs[1] shape on data that is NOT exact in fp16 (fp16 MFMA scan + error-bound guard)
          also["glove1.2m"] configs[2]: 1.2M x 50, inner product
          also.ivf1024      configs[3]: IVF-Flat nlist = 1024 over the sift1m data, ONE index trained by the library's
                            own k-means, nprobe 8 / 32 / 128
          also.msmarco_ivf  the reference's committed IVF-on-embeddings shape (100 000 x 384, cosine, IVF100,Flat,
                            nprobe 32, k = 20: benchmark_results/.../msmarco/ivf_flat_results.json) on the K-loop list scan
          also.serving      1 / 64-query batches: latency + the HBM roofline of the scan, on the Infinity-Cache-resident
                            1M x 128 copy (labelled so) and on a 4M x 128 copy (also.serving.bytes4m, 512 MB > 256 MiB: a true HBM number)
          also["marco12.5m"] configs[4] per-GPU shard, the N = 1 point of the N > 1 series
N > 1   the SAME headline workload, weak scaling: every rank holds its own 1M x 128 sift1m shard (rows seeded by rank, ids
        offset by rank x 1M), the same 10 000 queries; `value` = N x queries / time -- "query x shard scans per second" -- so the
        driver's N = 1 / 2 / 4 / 8 series is ONE workload and its N = 1 point is the BENCH line.  One process per GPU
        (torch.distributed, backend nccl = RCCL); launched by the driver's torchrun line, or by this script itself.
        Every rank scans ITS shard for the SAME query batch (vdb_search_partial_device); the packed partial (key64, id) lists
        are exchanged with ONE all-gather and merged on every rank; `exchange_ms` times that all-gather + merge alone.
        also["marco12.5m"] carries BASELINE.json configs[4] (12.5M x 768 inner product per GPU, rows generated on device,
        block seeds by GLOBAL block number: the corpus does not depend on N) with BOTH figures:
          weak    every rank holds 12.5M rows (100M at N = 8); value = N x queries / time
          strong  the 12.5M-row corpus is SPLIT over the N ranks (same rows, same queries, same result for every N:
                  `result_checksum`); value = queries / s against the whole corpus
        (--workload marco12.5m / --scaling strong put one of them in the headline instead.)
        Rehearsal on a one-GPU box: VDBHIP_BENCH_SHARED_GPU=1 (every rank on GPU 0, gloo collectives staged through the
        host): tests/test_gpu_bench_ranks.py runs this file's N = 2 code that way.

Output: ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
import zlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "vectordb-retrieval_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

PEAK_F16_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_I8_TOPS = 5000.0      # dense int8 MFMA peak (2x fp16 per clock), same guide
HBM_PEAK_GBS = 8000.0
WORKLOADS = {
    #  name        (rows, dim, queries, k, metric, generator)
    "sift1m": (1_000_000, 128, 10_000, 10, "l2", "sift_like"),
    "gaussian1m": (1_000_000, 128, 10_000, 10, "l2", "gaussian"),
    "glove1.2m": (1_200_000, 50, 10_000, 10, "ip", "glove_like"),
    "marco2m": (2_000_000, 768, 10_000, 10, "ip", "gaussian"),   # MS MARCO-shaped shard slice (config 5 is 12.5M/GPU)
    # BASELINE configs[4] per-GPU shard (100M x 768 over 8 GPUs): rows generated ON DEVICE in fixed 500k-row
    # blocks seeded by the global block number, so the data do not depend on the number of ranks
    "marco12.5m": (12_500_000, 768, 10_000, 10, "ip", "device_gaussian"),
    "marco1m": (1_000_000, 768, 2_000, 10, "ip", "device_gaussian"),       # the same shape in small: N > 1 rehearsals (tests)
    "gauss50m": (50_000_000, 128, 10_000, 10, "l2", "device_gaussian"),     # capacity check of the flat D <= 128 path
    "bytes4m": (4_000_000, 128, 64, 10, "l2", "device_bytes"),              # serving leg: scan copy > Infinity Cache
    "smoke": (10_000, 128, 100, 10, "l2", "random_reference"),
    # BASELINE configs[3]: IVF-Flat over the sift1m data, nlist = 1024, --nprobe 8 / 32 / 128
    "ivf1024": (1_000_000, 128, 10_000, 10, "l2", "sift_like"),
    # the reference's committed IVF-on-embeddings benchmark shape (msmarco subset): cosine = normalise + ip
    "msmarco_ivf": (100_000, 384, 10_000, 20, "ip", "embedding_like"),
}
DEVICE_BLOCK_ROWS = 500_000


def usable_cpus() -> int:
    """CPUs this process may use: the affinity mask, cut to the cgroup quota when there is one (a GPU box gives one
    GPU's share of a 256-thread host).  EVERY CPU leg runs on exactly this many threads and reports it as `cores`."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, n)


def device_rows_range(lo: int, hi: int, d: int, dev):
    """Rows [lo, hi) of the device-generated corpus: block b = rows [b * 500k, (b + 1) * 500k) from seed 1234 + b."""
    import torch

    X = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    for b in range(lo // DEVICE_BLOCK_ROWS, -(-hi // DEVICE_BLOCK_ROWS)):
        b0, b1 = b * DEVICE_BLOCK_ROWS, (b + 1) * DEVICE_BLOCK_ROWS
        gen.manual_seed(1234 + b)
        if b0 >= lo and b1 <= hi:
            X[b0 - lo:b1 - lo].normal_(generator=gen)
        else:            # a rank boundary inside the block: generate the whole block, keep the overlap
            tmp = torch.empty((DEVICE_BLOCK_ROWS, d), dtype=torch.float32, device=dev).normal_(generator=gen)
            s0, s1 = max(lo, b0), min(hi, b1)
            X[s0 - lo:s1 - lo] = tmp[s0 - b0:s1 - b0]
            del tmp
    return X


def device_byte_rows(n: int, d: int, dev, seed: int):
    """(n, d) float32 rows of integers 0..218, exponentially skewed like SIFT descriptors, generated on the device (the
    host generator of the sift1m rows needs 10 s per million)."""
    import torch

    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    X = torch.empty((n, d), dtype=torch.float32, device=dev)
    for lo in range(0, n, 1_000_000):
        X[lo:lo + 1_000_000].exponential_(1.0 / 28.0, generator=gen)
    return X.round_().clamp_(0.0, 218.0)


def device_rows(n: int, d: int, rank: int, dev):
    """(n, d) float32 standard-normal rows of rank `rank` under weak scaling: global rows [rank * n, (rank + 1) * n)."""
    blocks = -(-n // DEVICE_BLOCK_ROWS)
    base = rank * blocks * DEVICE_BLOCK_ROWS
    return device_rows_range(base, base + n, d, dev)


def device_check(X_t, q_t, I_t, k: int, metric: str, id_base: int, sample: int = 32, ranks=None) -> float:
    """Recall of the first `sample` queries against a float64 torch scan of the device-resident corpus (used for
    workloads too large to hand to the CPU oracle).  With world > 1 every rank scans ITS shard, the per-shard exact
    top-k (scores + global ids) are all-gathered and merged, and the merged list is what the result is compared with
    (the result holds neighbours from every shard)."""
    import torch

    q = q_t[:sample].double()
    best_v, best_i = None, None
    for lo in range(0, X_t.shape[0], 1_000_000):
        x = X_t[lo:lo + 1_000_000].double()
        s = q @ x.T if metric == "ip" else -((q * q).sum(1, keepdim=True) - 2.0 * (q @ x.T) + (x * x).sum(1)[None, :])
        v, i = torch.topk(s, k, dim=1)
        i = i + lo + id_base
        if best_v is None:
            best_v, best_i = v, i
        else:
            cv, ci = torch.cat([best_v, v], 1), torch.cat([best_i, i], 1)
            best_v, sel = torch.topk(cv, k, dim=1)
            best_i = torch.gather(ci, 1, sel)
    if ranks is not None and ranks.world > 1:
        vs, ids = ranks.all_gather_list(best_v), ranks.all_gather_list(best_i)
        cv, ci = torch.cat(vs, 1), torch.cat(ids, 1)
        best_v, sel = torch.topk(cv, k, dim=1)
        best_i = torch.gather(ci, 1, sel)
    got = I_t[:sample].cpu().numpy()
    ref = best_i.cpu().numpy()
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(ref, got)]))


def real_sift():
    """(X, Q, GT, where) from real SIFT1M TEXMEX files under $VDBHIP_DATA, or None (there are none offline)."""
    from vdbhip import io

    root = os.environ.get("VDBHIP_DATA")
    if not root:
        return None
    for d in (Path(root), Path(root) / "sift", Path(root) / "sift1m"):
        base, query, gt = d / "sift_base.fvecs", d / "sift_query.fvecs", d / "sift_groundtruth.ivecs"
        if base.exists() and query.exists():
            X, Q = io.read_fvecs(base), io.read_fvecs(query)
            G = io.read_ivecs(gt) if gt.exists() else None
            return np.ascontiguousarray(X), np.ascontiguousarray(Q), G, str(d)
    return None


def embedding_like(n: int, nq: int, d: int, seed_x: int, seed_q: int, latent: int = 32, topics: int = 64):
    """Sentence-embedding shaped unit vectors: the rows live near a `latent`-dimensional subspace of R^d (a fixed random
    map of a topic centre + latent Gaussian, plus a little isotropic noise), so neighbour distances spread the way they do
    in real embedding sets -- i.i.d. Gaussian rows in 384 dimensions would put every row at the same distance from every
    query (distance concentration), which no embedding model produces."""
    basis = np.random.default_rng(97).standard_normal((latent, d)).astype(np.float32) / np.sqrt(latent)
    centres = np.random.default_rng(98).standard_normal((topics, latent)).astype(np.float32)

    def rows(m, seed):
        rng = np.random.default_rng(seed)
        z = centres[rng.integers(0, topics, m)] + 0.8 * rng.standard_normal((m, latent), dtype=np.float32)
        x = z @ basis + 0.05 * rng.standard_normal((m, d), dtype=np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return np.ascontiguousarray(x, np.float32)

    return rows(n, seed_x), rows(nq, seed_q)


def make_data(name: str, rank: int):
    from vdbhip import datasets

    n, d, nq, k, metric, gen = WORKLOADS[name]
    if gen == "device_gaussian":
        Q = np.random.default_rng(1235).standard_normal((nq, d), dtype=np.float32)
        return None, Q, k, metric
    if gen == "sift_like":
        X = datasets._sift_rows(np.random.default_rng(1234 + 7919 * rank), n, d)
        Q = datasets._sift_rows(np.random.default_rng(1235), nq, d)
    elif gen == "gaussian":
        X = np.random.default_rng(1234 + 7919 * rank).standard_normal((n, d), dtype=np.float32)
        Q = np.random.default_rng(1235).standard_normal((nq, d), dtype=np.float32)
    elif gen == "glove_like":
        X = 0.5 * np.random.default_rng(50 + 7919 * rank).standard_normal((n, d), dtype=np.float32)
        Q = 0.5 * np.random.default_rng(51).standard_normal((nq, d), dtype=np.float32)
    elif gen == "embedding_like":
        X, Q = embedding_like(n, nq, d, 384 + 7919 * rank, 385)
    else:
        X, Q = datasets.random_reference(d, n, nq, 42)
    return np.ascontiguousarray(X, np.float32), np.ascontiguousarray(Q, np.float32), k, metric


def recall_vs(ids_ref, ids_got, k) -> float:
    return float(np.mean([len(set(a[:k].tolist()) & set(b[:k].tolist())) / k for a, b in zip(ids_ref, ids_got)]))


def blas_leg(X, Q, k, metric, budget_s):
    """Threaded-BLAS GEMM expansion with BLAS and select threads both limited to usable_cpus()."""
    from oracle import blas_baseline

    cpus = usable_cpus()
    try:
        from threadpoolctl import threadpool_limits

        ctx = threadpool_limits(limits=cpus, user_api="blas")
    except Exception:  # noqa: BLE001
        import contextlib

        ctx = contextlib.nullcontext()
    with ctx:
        leg, ids = blas_baseline.time_gemm_expansion(X, Q, k, metric, budget_s=budget_s, threads=cpus)
    return leg, ids


def c_port_leg(X, Q, k, metric, budget_s):
    """The hand-written C port (scalar / omp simd dot loops, no BLAS) on usable_cpus() OpenMP threads."""
    from oracle import c_oracle

    c_oracle.build()
    cores = usable_cpus()
    probe = min(len(Q), 4 * cores)
    t0 = time.perf_counter()
    c_oracle.knn(X, Q[:probe], k, metric, mode=c_oracle.MODE_GEMM32, threads=cores)
    dt = time.perf_counter() - t0
    sample = int(min(len(Q), max(probe, probe * budget_s / max(dt, 1e-6))))
    t0 = time.perf_counter()
    c_oracle.knn(X, Q[:sample], k, metric, mode=c_oracle.MODE_GEMM32, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(sample / dt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
            "impl": f"oracle/knn_oracle.c MODE_GEMM32 (float32 expansion, OpenMP on {cores} threads, no BLAS)",
            "sample": f"first {sample} of {len(Q)} queries against all {len(X)} rows, {dt:.1f} s"}


def timed_device_loop(index, q_t, nq, k, D_t, I_t, stream, steps, warmup, torch):
    for _ in range(warmup):
        index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
    torch.cuda.synchronize()
    index.set_option("timing", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = index.stats()
    index.set_option("timing", 0)
    return elapsed, st


def host_io_timing(algo, Q, k, first=True):
    """The reference harness's view (experiment_runner.py:330, 428-439, 464): batch_search timed with time.time(),
    pageable NumPy queries in, NumPy (D, I) out; the very first call after build_index, then the median of 11."""
    out = {}
    if first:
        t1 = time.time()
        algo.batch_search(Q, k=k)
        out["first_call_ms"] = round((time.time() - t1) * 1e3, 3)
    for _ in range(3):
        algo.batch_search(Q, k=k)
    times = []
    for _ in range(11):
        t1 = time.time()
        algo.batch_search(Q, k=k)
        times.append(time.time() - t1)
    med = float(np.median(times))
    out["value_host_io"] = round(len(Q) / med, 1)
    out["host_io_ms"] = round(med * 1e3, 4)
    return out


HOST_IO_NOTE = ("value_host_io = SURVEY 8(d): wall clock of <plugin>.batch_search(Q pageable numpy) -> numpy (D, I), H2D of Q "
                "and D2H of the result inside the timed call (experiment_runner.py:428-439), median of 11 after 3 "
                "warm-ups; first_call_ms = the very first call after build_index (the reference has no warm-up; "
                "build_index sizes the workspace for 10 000 queries: vdb_reserve); `value` = the same batch with the "
                "queries and the result resident in HBM, K steps between two synchronisations")


def pipeline_of(st, nq, build_s, corpus_bytes):
    res = float(st["bytes_resident"])
    ws = float(st.get("bytes_workspace", 0))
    return {"path": st["last_path_name"], "candidates_per_query": round(st["last_candidates"] / nq, 2),
            "hbm_index_mb": round((res - ws) / 2 ** 20, 1), "hbm_workspace_mb": round(ws / 2 ** 20, 1),
            "hbm_index_over_corpus": round((res - ws) / max(corpus_bytes, 1), 3),
            "layout": {0: "float32 rows + fp16 scan copy", 1: "float32 rows + fp16 and int8 scan copies",
                       2: "int8 rows + int8 scan copy only (option int8_only)"}.get(int(st.get("has_i8_copy", 0)), "?"),
            "rescan_bins": int(st["last_rescan_bins"]), "fallback_queries": int(st["last_fallback_queries"]),
            "corpus_fp16_exact": int(st["corpus_fp16_exact"]), "build_s": round(build_s, 3),
            "hbm_resident_mb": round(res / 2 ** 20, 1), "corpus_mb": round(corpus_bytes / 2 ** 20, 1),
            "hbm_resident_over_corpus": round(res / max(corpus_bytes, 1), 3)}


def scan_dtype_name(st) -> str:
    return ("i8 MFMA scan (i32 accumulate) + exact integer / f64 refine" if int(st.get("scan_dtype", 0)) == 1
            else "f16 MFMA scan (f32 accumulate) + f64 exact refine")


def flat_leg(vdbhip, torch, name, dev, local_rank, stream, steps, warmup, cpu_budget_s=6.0, host_io=True,
             data=None, keep_index=False, all_cpu_legs=False, engine_options=None):
    """One brute-force workload through the plugin: build, host-I/O timing, device-resident timed loop with the scan's
    HIP-event roofline, recall against a CPU leg on a bounded query sample."""
    n, d = WORKLOADS[name][:2]
    X, Q, k, metric = data if data is not None else make_data(name, 0)
    n, d = X.shape
    nq = Q.shape[0]
    t0 = time.perf_counter()
    extra = {"engine_options": dict(engine_options)} if engine_options else {}
    algo = vdbhip.get_algorithm_instance("HipExactSearch", d, name="bench", metric=metric, device=local_rank, **extra)
    algo.build_index(X)
    build_s = time.perf_counter() - t0
    leg = {"engine_options": dict(engine_options)} if engine_options else {}
    if host_io:
        leg.update(host_io_timing(algo, Q, k))
    index = algo.index
    q_t = torch.from_numpy(Q).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    el, st = timed_device_loop(index, q_t, nq, k, D_t, I_t, stream, steps, warmup, torch)
    ids = I_t.cpu().numpy()
    out = {"value": round(nq * steps / el, 1), "unit": "queries/s", "ms_per_step": round(el / steps * 1e3, 4),
           "config": f"{name}: {n} rows x {d} dims, {nq} queries, k={k}, {metric}; brute-force exact k-NN",
           "dtype": scan_dtype_name(st)}
    out.update(leg)
    out["roofline"] = roofline_of(st, nq, n, d, name)
    out["pipeline"] = pipeline_of(st, nq, build_s, X.nbytes)
    if cpu_budget_s > 0:
        bl, bids = blas_leg(X, Q, k, metric, cpu_budget_s)
        out["cpu_baseline"] = bl
        out["recall@10_vs_cpu_blas_sample"] = round(recall_vs(bids, ids[:len(bids)], k), 6)
        if all_cpu_legs:
            from oracle import blas_baseline

            legs = {"blas_gemm_expansion": bl, "c_port": c_port_leg(X, Q, k, metric, cpu_budget_s * 0.6)}
            if metric == "l2":   # the reference's own CPU path for YAML `exact` (NumPy LinearSearcher), tiny query batches
                legs["numpy_linear_searcher"] = blas_baseline.time_linear_searcher(X, Q, k, metric, qbatch=4, budget_s=4.0)
            out["cpu_baselines"] = legs
        from oracle import c_oracle

        _, io_ = c_oracle.knn(X, Q[:32], k, metric, threads=usable_cpus())
        out["ids_equal_cpu_oracle_first32"] = bool(np.array_equal(ids[:32], io_))
    if keep_index:
        return out, (algo, index, q_t, D_t, I_t, X, Q, ids)
    index.close()
    return out


def ivf_leg(vdbhip, torch, dev, local_rank, stream, steps, warmup, nprobes, name="ivf1024", nlist=1024, data=None,
            cpu_budget_s=5.0, plugin_metric=None):
    """BASELINE configs[3]: ONE IVF-Flat index (the library's own k-means: vdb_ivf_train 25 iterations, seed 1234),
    `set_nprobe` per point; each point with the list scan's roofline, recall@10 against the exact result of the same
    queries, and the C restatement of the same search (same centroids, same lists) as the CPU leg and bit-level check."""
    from oracle import c_oracle

    X, Q, k, metric = data if data is not None else make_data(name, 0)
    n, d = X.shape
    nq = Q.shape[0]
    t0 = time.perf_counter()
    index = vdbhip.IVFFlatIndex(d, nlist, metric, local_rank)
    index.train(X, niter=25, seed=1234, max_points_per_centroid=256)
    train_s = time.perf_counter() - t0
    index.add(X)
    build_s = time.perf_counter() - t0
    flat = vdbhip.FlatIndex(d, metric, local_rank)
    flat.add(X)
    _, exact_ids = flat.search(Q, k)
    flat.close()
    C, lor = index.centroids(), index.assignment()
    sizes = np.bincount(lor, minlength=nlist)
    q_t = torch.from_numpy(Q).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    c_oracle.build()
    cores = usable_cpus()
    out = {"config": f"{name}: {n} rows x {d} dims, {nq} queries, k={k}, {metric}; IVF-Flat nlist={nlist}, own k-means "
                     f"(25 iterations on {min(n, 256 * nlist)} sampled rows)",
           "train_s": round(train_s, 3), "build_s": round(build_s, 3),
           "list_rows": {"min": int(sizes.min()), "median": int(np.median(sizes)), "max": int(sizes.max())}}
    for p in nprobes:
        index.set_nprobe(p)
        host = {}
        for _ in range(2):
            index.search(Q, k)
        ts = []
        for _ in range(7):
            t1 = time.time()
            index.search(Q, k)
            ts.append(time.time() - t1)
        host["value_host_io"] = round(nq / float(np.median(ts)), 1)
        el, st = timed_device_loop(index, q_t, nq, k, D_t, I_t, stream, steps, warmup, torch)
        ids = I_t.cpu().numpy()
        rows_probed = float(st.get("last_rows_scanned", 0)) or nq * p / float(nlist) * n
        roof = roofline_of(st, nq, n, d, name, ivf_rows_probed=rows_probed, traffic_key=f"{name}_nprobe{p}")
        roof["bytes_avoided"] = {"per_query_scan_bytes": 4.0 * d * rows_probed,
                                 "note": "SURVEY 8(d)'s per-query IVF figure (4*D bytes x (query, row) pairs): what a query-major "
                                         "list scan would stream per batch.  NOT a roofline: the list-major scan reads every "
                                         "probed list once per group of query slots, so it is MFMA- (nprobe 32 / 128) or "
                                         "launch-bound (nprobe 8), not bound by these bytes"}
        point = {"value": round(nq * steps / el, 1), "unit": "queries/s", "ms_per_step": round(el / steps * 1e3, 4),
                 "dtype": scan_dtype_name(st), **host, "roofline": roof,
                 "recall@10_vs_exact": round(recall_vs(exact_ids, ids, min(k, 10)), 6),
                 "pipeline": {"path": st["last_path_name"], "candidates_per_query": round(st["last_candidates"] / nq, 2),
                              "rescan_bins": int(st["last_rescan_bins"]),
                              "fallback_queries": int(st["last_fallback_queries"]),
                              "rows_scanned_per_query": round(rows_probed / nq, 1),
                              "hbm_resident_mb": round(st["bytes_resident"] / 2 ** 20, 1)}}
        if k != 10:
            point[f"recall@{k}_vs_exact"] = round(recall_vs(exact_ids, ids, k), 6)
        if cpu_budget_s > 0:
            probe = min(nq, 128)
            t1 = time.perf_counter()
            c_oracle.ivf_search(X, C, lor, Q[:probe], k, p, metric, threads=cores)
            dt = time.perf_counter() - t1
            sample = int(min(nq, max(probe, probe * cpu_budget_s / max(dt, 1e-6))))
            t1 = time.perf_counter()
            _, io_ = c_oracle.ivf_search(X, C, lor, Q[:sample], k, p, metric, threads=cores)
            dt = time.perf_counter() - t1
            point["cpu_baseline"] = {"value": round(sample / dt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                                     "impl": f"oracle/ivf_oracle.c (the same IVF-Flat search: same centroids, same lists, "
                                             f"canonical float64 list scan, OpenMP on {cores} threads); FAISS is not "
                                             f"installed on the box",
                                     "sample": f"first {sample} of {nq} queries, nprobe {p}, {dt:.1f} s"}
            point["ids_equal_cpu_oracle_sample"] = bool(np.array_equal(io_, ids[:sample]))
        out[f"nprobe{p}"] = point
    index.close()
    return out


def scaling_reference_leg(vdbhip, torch, dev, local_rank, stream, steps, warmup):
    """The N > 1 default workload (one config-5 shard, 12.5M x 768 inner product) on ONE GPU through the sharded code
    path (partial lists -> packed buffer -> merge; the all-gather of a one-rank world is a copy): the N = 1 point of
    the weak- and strong-scaling series, in the same run as the headline."""
    name = "marco12.5m"
    n, d, nq, k, metric, _ = WORKLOADS[name]
    _, Q, _, _ = make_data(name, 0)
    X_t = device_rows(n, d, 0, dev)
    t0 = time.perf_counter()
    index = vdbhip.FlatIndex(d, metric, local_rank)
    index.add_device(X_t.data_ptr(), n, id_base=0)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    q_t = torch.from_numpy(Q).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    my_pack = torch.empty((2, nq, k), dtype=torch.int64, device=dev)
    all_pack = torch.empty((1, 2, nq, k), dtype=torch.int64, device=dev)

    def step():
        index.search_partial_device(q_t.data_ptr(), nq, k, my_pack[0].data_ptr(), my_pack[1].data_ptr(), stream)
        all_pack.copy_(my_pack.unsqueeze(0))
        vdbhip.merge_packed_partials_device(metric, local_rank, all_pack.data_ptr(), 1, nq, k,
                                            D_t.data_ptr(), I_t.data_ptr(), stream)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    index.set_option("timing", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = index.stats()
    leg = {"value": round(nq * steps / el, 1), "unit": "queries/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(el / steps * 1e3, 4),
           "config": f"{name}: {n} rows x {d} dims, {nq} queries, k={k}, {metric}; rows generated on device",
           "roofline": roofline_of(st, nq, n, d, name),
           "pipeline": pipeline_of(st, nq, build_s, float(n) * d * 4),
           "recall@10_vs_float64_torch_sample": round(device_check(X_t, q_t, I_t, k, metric, 0), 6),
           "result_checksum": result_checksum(I_t),
           "note": "N = 1 point of both N > 1 series: `bench.py --gpus N` runs this shard on every GPU (weak: value = N x "
                   "queries / time, efficiency = value(N) / (N x this value)); `--scaling strong` splits these 12.5M rows "
                   "over the N GPUs (speed-up = value(N) / this value, and result_checksum must not change)"}
    index.close()
    return leg


def serving_leg(vdbhip, torch, dev, local_rank, stream, index, q_t, k, n, d, tag):
    """Serving-shaped batches: wall time of one search_device + synchronise (median of 30) for 1 and 64 queries, and the
    HBM roofline of the scan -- a single query streams the whole scan copy once."""
    leg = {}
    D_t = torch.empty((64, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((64, k), dtype=torch.int64, device=dev)
    for nq in (1, 64):
        for _ in range(5):
            index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):                     # wall latency, without the library's event records in the stream
            t0 = time.perf_counter()
            index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        index.set_option("timing", 1)           # ... and the scan's own time from HIP events on the search stream
        for _ in range(30):
            index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
        torch.cuda.synchronize()
        st = index.stats()
        index.set_option("timing", 0)
        i8 = int(st.get("scan_dtype", 0)) == 1
        dpad = -(-d // 32) * 32 if i8 else -(-d // 16) * 16
        panel_bytes = float(n) * dpad * (1 if i8 else 2)
        scan_ms = float(st["last_scan_ms"])
        traffic, source = None, None
        try:
            ent = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get(tag + "_serving_nq1", {})
            traffic, source = ent.get("hbm_bytes_per_launch"), ent.get("source")
        except Exception:  # noqa: BLE001
            pass
        leg[f"nq{nq}"] = {"latency_us": round(float(np.median(ts)) * 1e6, 1), "scan_us": round(scan_ms * 1e3, 1),
                          "roofline": {"bound": "hbm", "achieved": round(panel_bytes / (scan_ms * 1e-3) / 1e9, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(panel_bytes / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "algorithmic_bytes_per_launch": panel_bytes, "traffic": traffic,
                                       "traffic_source": (source + " -- a recorded PMC pass, not measured in this run") if source else None,
                                       "note": "rows x padded dims x %d B: the %s scan copy read once" %
                                               (1 if i8 else 2, "int8" if i8 else "fp16")}}
    copy_mb = leg["nq1"]["roofline"]["algorithmic_bytes_per_launch"] / 2 ** 20
    leg["scan_copy_mb"] = round(copy_mb, 1)
    leg["residency"] = ("Infinity-Cache resident: the scan copy fits the 256 MiB MALL, repeated single-query scans need not "
                        "touch HBM -- the GB/s here is a cache number, NOT an HBM roofline fraction" if copy_mb <= 256.0 else
                        "HBM: the scan copy is larger than the 256 MiB Infinity Cache, every scan streams it from HBM")
    return leg


def roofline_of(st, nq, n, d, workload, ivf_rows_probed=None, traffic_key=None):
    """MFMA roofline of the dominant kernel from the HIP-event time the library recorded on the search stream."""
    scan_ms = float(st["last_scan_ms"])
    i8 = int(st.get("scan_dtype", 0)) == 1
    rows = float(n) if ivf_rows_probed is None else float(ivf_rows_probed)
    flops = 2.0 * nq * rows * d if ivf_rows_probed is None else 2.0 * rows * d
    peak = PEAK_I8_TOPS if i8 else PEAK_F16_TFLOPS
    achieved = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    if ivf_rows_probed is not None:
        if d > 128:
            kernel = "scan16_kloop_kernel<ITEMS> (list-major IVF scan, K-loop)"
        else:
            kernel = ("scan_i8_kernel" if i8 else "scan_kernel") + "<ITEMS> (list-major IVF scan)"
    elif d <= 128 and int(st.get("scan_shape", 32)) == 16:      # layout "x16": 16x16x64 i8 / 16x16x32 f16 MFMA
        # (an index with an int8 copy runs both scans from ONE launch, scan_pair_x16_kernel: the device picks the body)
        pair = int(st.get("has_i8_copy", 0)) == 1
        kernel = (("scan_pair_x16_kernel -> " if pair else "") +
                  ("scan_i8x16_%s<%d>" % ("body" if pair else "kernel", 1 if d <= 64 else 2) if i8 else
                   "scan_x16_%s<%d>" % ("body" if pair else "kernel", 2 if d <= 64 else 4)))
    elif d <= 128:
        kernel = "scan_i8_kernel" if i8 else "scan_kernel<%d>" % (4 if d <= 64 else 8)
    else:
        kernel = "scan16_kloop_kernel"
    traffic, source = None, None
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    if pmc.exists():
        try:
            ent = json.loads(pmc.read_text()).get(traffic_key or (workload + ("_i8" if i8 else "")), {})
            traffic, source = ent.get("hbm_bytes_per_launch"), ent.get("source")
        except Exception:  # noqa: BLE001
            pass
    memory_side = {}
    if traffic and scan_ms > 0:     # what the recorded traffic amounts to at THIS run's kernel time (fabric-side bytes: HBM + Infinity Cache)
        rate = traffic / (scan_ms * 1e-3) / 1e12
        memory_side = {"traffic_tb_s": round(rate, 2), "traffic_over_hbm_peak": round(rate / (HBM_PEAK_GBS / 1000.0), 3)}
        if ivf_rows_probed is not None:
            # the list-major scans gather their B fragments (64-byte segments of the query rows: per K-step for D > 128, per work
            # item for D <= 128) and re-read list panels per slot group, all from the Infinity Cache: at nprobe 32 / 128 their
            # binding resource is that gather rate, not the matrix pipe (PMC: 5.0 - 5.3 TB/s fabric-side at both widths)
            memory_side["second_bound"] = {"bound": "infinity-cache gather", "achieved": round(rate, 2), "peak": 8.6, "unit": "TB/s",
                                           "frac": round(rate / 8.6, 3),
                                           "note": "peak = gathered rows from an Infinity-Cache-resident table, MI355X_MICROARCH.md "
                                                   "'Indexed rows: gather into LDS' (8.6 TB/s chip-wide)"}
    # the three stages of a search from the library's HIP events (vdb_stats: prep | dominant kernel | tail): the roofline names
    # the scan, and says so when another stage is the longest (the msmarco-shaped leg's tail outlasted its scan)
    pipeline_ms = float(st["last_total_ms"])
    stages = {"prep": round(float(st.get("last_prep_ms", 0.0)), 4), "scan": round(scan_ms, 4),
              "tail": round(float(st.get("last_tail_ms", 0.0)), 4)}
    longest = max(stages, key=stages.get)
    stage_names = {"prep": ("query statistics + operands" + (" + coarse quantizer search + plan" if ivf_rows_probed is not None else "")),
                   "scan": kernel,
                   "tail": ("ivf_select_kernel + ivf_tail_kernel" if ivf_rows_probed is not None else "select_kernel + refine_tail_kernel")
                           + " (bin select + exact refine)"}
    return {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": peak,
            "unit": "TOP/s (int8)" if i8 else "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, **memory_side,
            "traffic_source": (source + " -- a recorded PMC pass of an earlier run of this command, not measured in "
                               "this run") if source else None,
            "kernel_ms": round(scan_ms, 4), "pipeline_ms": round(pipeline_ms, 4),
            "stages_ms": stages, "longest_stage": longest, "longest_stage_kernels": stage_names[longest],
            "scan_share_of_pipeline": round(scan_ms / pipeline_ms, 4) if pipeline_ms > 0 else None,
            "end_to_end_frac": round(flops / (pipeline_ms * 1e-3) / 1e12 / peak, 4) if pipeline_ms > 0 else None,
            "algorithmic_flops_per_launch": flops}


def result_checksum(I_t) -> str:
    """CRC32 of the (nq, k) int64 result ids: equal across N under --scaling strong (the merge is shard-count invariant)."""
    return "%08x" % (zlib.crc32(I_t.cpu().numpy().tobytes()) & 0xFFFFFFFF)


def shared_gpu() -> bool:
    """Rehearsal mode ($VDBHIP_BENCH_SHARED_GPU=1): every rank uses GPU 0 and the collectives run over gloo, staged through the
    host (RCCL refuses two ranks on one device) -- the N > 1 code of this file on a one-GPU box (tests/test_gpu_bench_ranks.py)."""
    return os.environ.get("VDBHIP_BENCH_SHARED_GPU") == "1"


def launch_ranks(args) -> int:
    """--gpus N > 1 without a torchrun environment: this parent starts one child per GPU BEFORE it makes any GPU call
    (counting devices does not initialise one), relays rank 0's JSON line and fails if any rank fails.  Every child is
    polled: the first one to exit non-zero takes its siblings down with it (they would otherwise sit in the RCCL
    rendezvous or a collective until its timeout)."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and not shared_gpu():
        raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {have}")
    port = int(os.environ.get("MASTER_PORT", 29400 + os.getpid() % 2000))
    procs, out0 = [], ROOT / "gpurun_out" / f".bench_rank0_{os.getpid()}.out"
    out0.parent.mkdir(exist_ok=True)
    with open(out0, "wb") as f0:
        for rank in range(args.gpus):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                          stdout=f0 if rank == 0 else sys.stderr))
        rcs = wait_all_or_kill(procs)
    text = out0.read_text(errors="replace")
    out0.unlink(missing_ok=True)
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return 1
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 1
    print(lines[-1], flush=True)
    return 0


def wait_all_or_kill(procs, poll_s: float = 0.2, grace_s: float = 5.0):
    """Exit codes of `procs`; as soon as one exits non-zero the others are terminated (then killed after grace_s)."""
    rcs = [None] * len(procs)
    failed = False
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
                if rcs[i] not in (None, 0):
                    failed = True
        if failed:
            deadline = time.time() + grace_s
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
            break
        time.sleep(poll_s)
    return rcs


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--nprobe", type=int, default=0, help="IVF workloads: lists probed per query (0 = the config's set)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N > 1: weak = every rank holds the workload's rows; strong = the rows are split over the ranks")
    ap.add_argument("--stream-panels", action="store_true",
                    help="device-generated D > 128 workloads: option stream_panels = 1 (the fp16 scan copy is converted per "
                         "search instead of kept resident: smaller footprint, one extra pass over the rows per batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline workload only: no also.* legs")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    # the headline workload is the SAME for every N (BASELINE.json configs[1]): sift1m, weak scaling for N > 1 -- every rank holds
    # its own 1M x 128 shard, value = N x queries / time, so the N = 1 point of the driver's scaling series IS the BENCH line
    workload = args.workload or "sift1m"

    # stdout carries exactly ONE line, the JSON result: everything native libraries print on file descriptor 1 while
    # the job runs (the pool exports NCCL_DEBUG=VERSION, so RCCL prints a five-line banner there) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if shared_gpu() else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch

    import vdbhip
    from vdbhip import _ffi

    if not torch.cuda.is_available() or _ffi.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    cpu_budget = 0.0 if args.no_cpu_baseline else 6.0

    if world == 1 and os.environ.get("VDBHIP_BENCH_FORCE_SHARDED") != "1":
        out = single_gpu_line(args, workload, vdbhip, torch, dev, local_rank, stream, cpu_budget)
    else:
        out = sharded_line(args, workload, vdbhip, torch, dev, rank, local_rank, world, stream)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0


def single_gpu_line(args, workload, vdbhip, torch, dev, local_rank, stream, cpu_budget):
    n, d, nq, k, metric, gen = WORKLOADS[workload]
    contract = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "rccl_ranks": 1}
    if workload in ("ivf1024", "msmarco_ivf"):       # an IVF workload on its own (profiling runs): one point is the line
        nlist = 1024 if workload == "ivf1024" else 100
        probes = [args.nprobe] if args.nprobe > 0 else ([8, 32, 128] if workload == "ivf1024" else [32])
        leg = ivf_leg(vdbhip, torch, dev, local_rank, stream, args.steps, args.warmup, probes, name=workload,
                      nlist=nlist, cpu_budget_s=cpu_budget)
        head = leg[f"nprobe{probes[-1]}"]
        out = {"metric": f"QPS ({workload} IVF-Flat nlist={nlist} nprobe={probes[-1]}, k={k})", "value": head["value"],
               "unit": "queries/s", **contract, "ms_per_step": head["ms_per_step"], "dtype": head["dtype"],
               "data": "synthetic",
               "config": {"workload": leg["config"] + "; inputs resident in HBM", "rows_per_gpu": n, "dim": d,
                          "queries": nq, "k": k, "metric": metric, "nlist": nlist, "nprobe": probes[-1]},
               "roofline": head["roofline"], "ivf": leg}
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        return out
    if gen == "device_gaussian":
        leg = device_corpus_line(args, workload, vdbhip, torch, dev, local_rank, stream)
        leg.update(contract)
        return leg

    data_tag, GT, data = "synthetic", None, None
    real = real_sift() if workload == "sift1m" else None
    if real is not None:
        Xr, Qr, GT, where = real
        data = (Xr, Qr, 10, "l2")
        data_tag = f"sift1m (real TEXMEX files from {where})"
    head, kept = flat_leg(vdbhip, torch, workload, dev, local_rank, stream, args.steps, args.warmup,
                          cpu_budget_s=cpu_budget * 1.6, data=data, keep_index=True, all_cpu_legs=(workload == "sift1m"))
    algo, index, q_t, D_t, I_t, X, Q, ids = kept
    n, d = X.shape
    name = {"sift1m": "SIFT1M%s, 10k-query batch, k=10" % ("" if real else "-shaped")}.get(workload, f"{workload}, k={k}")
    out = {"metric": f"QPS @ recall@10 ({name})", "value": head["value"], "unit": "queries/s", **contract,
           "ms_per_step": head["ms_per_step"], "dtype": head["dtype"], "data": data_tag,
           "config": {"workload": head["config"] + "; `value` = inputs and results resident in HBM (bench contract)"
                                  + ("; with NumPy queries in and NumPy (D, I) out per SURVEY 8(d): %.2f M QPS, first call %.2f ms"
                                     % (head["value_host_io"] / 1e6, head["first_call_ms"]) if "value_host_io" in head else ""),
                      "rows_per_gpu": n, "dim": d, "queries": nq,
                      "k": k, "metric": metric, "sharding": "none"},
           "value_device_resident": head["value"]}
    for key in ("value_host_io", "host_io_ms", "first_call_ms"):
        if key in head:
            out[key] = head[key]
    out["value_note"] = HOST_IO_NOTE
    for key in ("roofline", "pipeline", "cpu_baseline", "cpu_baselines", "recall@10_vs_cpu_blas_sample",
                "ids_equal_cpu_oracle_first32"):
        if key in head:
            out[key] = head[key]
    if int(head["roofline"]["peak"]) == int(PEAK_I8_TOPS):
        out["dtype_note"] = ("the int8 scan serves byte-valued corpora with integer queries only (SIFT descriptors are uint8); "
                             "one non-integer query value puts the batch on the fp16 scan of the same index: "
                             "also.gaussian1m is that case")
    if GT is not None:
        out["recall@10_vs_sift_groundtruth"] = round(recall_vs(GT[:, :k], ids, k), 6)
    if args.no_extras or workload != "sift1m":
        index.close()
        return out

    also = {}
    also["serving"] = {"sift1m": serving_leg(vdbhip, torch, dev, local_rank, stream, index, q_t, k, n, d, "sift1m")}
    index.close()
    # the same corpus and queries on an int8-only index (engine option `int8_only`: the reference holds ONE copy of the corpus,
    # exact_search.py:34-39): footprint and rate, ids against the headline's
    lean, lkept = flat_leg(vdbhip, torch, workload, dev, local_rank, stream, args.steps, args.warmup, cpu_budget_s=0.0,
                           data=(X, Q, k, metric), keep_index=True, engine_options={"int8_only": 1})
    lean["ids_equal_default_index"] = bool(np.array_equal(lkept[7], ids))
    lkept[1].close()
    del lkept
    also["sift1m_int8_only"] = lean
    del X, Q, kept, algo
    also["gaussian1m"] = flat_leg(vdbhip, torch, "gaussian1m", dev, local_rank, stream, args.steps, args.warmup,
                                  cpu_budget_s=cpu_budget * 0.7)
    also["glove1.2m"] = flat_leg(vdbhip, torch, "glove1.2m", dev, local_rank, stream, args.steps, args.warmup,
                                 cpu_budget_s=cpu_budget)
    also["ivf1024"] = ivf_leg(vdbhip, torch, dev, local_rank, stream, args.steps, args.warmup, [8, 32, 128],
                              cpu_budget_s=cpu_budget * 0.8)
    also["msmarco_ivf"] = ivf_leg(vdbhip, torch, dev, local_rank, stream, args.steps, args.warmup, [32],
                                  name="msmarco_ivf", nlist=100, cpu_budget_s=cpu_budget * 0.8)
    # serving on a scan copy larger than the Infinity Cache: 4M x 128 byte-valued rows = 512 MB of int8 panels
    n4, d4, _, k4, m4, _ = WORKLOADS["bytes4m"]
    X4 = device_byte_rows(n4, d4, dev, 77)
    i4 = vdbhip.FlatIndex(d4, m4, local_rank)
    i4.add_device(X4.data_ptr(), n4, id_base=0)
    del X4
    torch.cuda.empty_cache()
    q4_t = device_byte_rows(64, d4, dev, 78)
    also["serving"]["bytes4m"] = serving_leg(vdbhip, torch, dev, local_rank, stream, i4, q4_t, k4, n4, d4, "bytes4m")
    i4.close()
    torch.cuda.empty_cache()
    also["marco12.5m"] = scaling_reference_leg(vdbhip, torch, dev, local_rank, stream, min(args.steps, 10),
                                               min(args.warmup, 2))
    out["also"] = also
    return out


def device_corpus_line(args, workload, vdbhip, torch, dev, local_rank, stream):
    """A device-generated workload (marco12.5m, gauss50m) on one GPU, plain (non-sharded) search path."""
    n, d, nq, k, metric, _ = WORKLOADS[workload]
    _, Q, _, _ = make_data(workload, 0)
    X_t = device_rows(n, d, 0, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = vdbhip.FlatIndex(d, metric, local_rank)
    if args.stream_panels:
        index.set_option("stream_panels", 1)
    index.add_device(X_t.data_ptr(), n, id_base=0)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    q_t = torch.from_numpy(Q).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    el, st = timed_device_loop(index, q_t, nq, k, D_t, I_t, stream, args.steps, args.warmup, torch)
    out = {"metric": f"QPS ({workload}, k={k})", "value": round(nq * args.steps / el, 1), "unit": "queries/s",
           "ms_per_step": round(el / args.steps * 1e3, 4), "dtype": scan_dtype_name(st), "data": "synthetic",
           "config": {"workload": f"{workload}: {n} rows x {d} dims per GPU, {nq} queries, k={k}, {metric}; brute-force "
                                  f"exact k-NN; rows generated on device; inputs resident in HBM",
                      "rows_per_gpu": n, "dim": d, "queries": nq, "k": k, "metric": metric, "sharding": "none"},
           "roofline": roofline_of(st, nq, n, d, workload),
           "pipeline": pipeline_of(st, nq, build_s, float(n) * d * 4),
           "recall@10_vs_float64_torch_sample": round(device_check(X_t, q_t, I_t, k, metric, 0), 6),
           "result_checksum": result_checksum(I_t), "stream_panels": bool(args.stream_panels)}
    index.close()
    return out


class Ranks:
    """The job's process group as bench.py uses it: RCCL (backend nccl, device tensors) on a multi-GPU node; in the shared-GPU
    rehearsal gloo with every collective staged through the host."""

    def __init__(self, torch, dev, world):
        self.torch, self.dev, self.world, self.dist = torch, dev, world, None
        self.backend = "none"
        if world > 1:
            import torch.distributed as dist

            self.dist = dist
            self.backend = os.environ.get("VDBHIP_BENCH_BACKEND") or ("gloo" if shared_gpu() else "nccl")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(self.backend)

    def all_gather_packed(self, all_pack, my_pack):
        """(2, nq, k) per rank -> (world, 2, nq, k): ONE collective."""
        if self.world == 1:
            all_pack.copy_(my_pack.unsqueeze(0))
        elif self.backend == "nccl":
            self.dist.all_gather_into_tensor(all_pack, my_pack)
        else:
            host = my_pack.cpu()
            parts = [self.torch.empty_like(host) for _ in range(self.world)]
            self.dist.all_gather(parts, host)
            all_pack.copy_(self.torch.stack(parts))

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather_list(self, t):
        """every rank's copy of the device tensor `t` (device_check)."""
        if self.backend == "nccl":
            out = [self.torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(out, t.contiguous())
            return out
        host = t.contiguous().cpu()
        parts = [self.torch.empty_like(host) for _ in range(self.world)]
        self.dist.all_gather(parts, host)
        return [p.to(t.device) for p in parts]

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def sharded_leg(args, workload, strong, vdbhip, torch, dev, rank, local_rank, ranks, stream, steps, warmup):
    """One row-sharded workload over the job's ranks: every rank scans ITS shard for the same query batch
    (vdb_search_partial_device), ONE all-gather of the packed partials, merge on every rank."""
    world = ranks.world
    n_total, d, nq, k, metric, gen = WORKLOADS[workload]
    if strong:
        if gen != "device_gaussian":
            raise SystemExit("--scaling strong needs a device-generated workload (marco12.5m, marco1m)")
        lo, hi = n_total * rank // world, n_total * (rank + 1) // world
    else:
        lo, hi = rank * n_total, (rank + 1) * n_total      # (global row ids; host-generated shards are seeded by rank)
    n = hi - lo
    if gen == "device_gaussian":
        _, Q, _, _ = make_data(workload, 0)
        X_t = device_rows_range(lo, hi, d, dev) if strong else device_rows(n, d, rank, dev)
        X = None
    else:
        X, Q, k, metric = make_data(workload, rank)
        X_t = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = vdbhip.FlatIndex(d, metric, local_rank)
    if args.stream_panels:
        index.set_option("stream_panels", 1)
    if X is None:
        index.add_device(X_t.data_ptr(), n, id_base=lo)
    else:
        index.add(X, id_base=lo)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0

    q_t = torch.from_numpy(Q).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    # one packed buffer per rank: keys (nq*k float64) immediately followed by ids (nq*k int64) -> ONE all-gather
    my_pack = torch.empty((2, nq, k), dtype=torch.int64, device=dev)
    all_pack = torch.empty((world, 2, nq, k), dtype=torch.int64, device=dev)

    def local_step():
        index.search_partial_device(q_t.data_ptr(), nq, k, my_pack[0].data_ptr(), my_pack[1].data_ptr(), stream)

    def exchange_step():
        ranks.all_gather_packed(all_pack, my_pack)
        vdbhip.merge_packed_partials_device(metric, local_rank, all_pack.data_ptr(), world, nq, k,
                                            D_t.data_ptr(), I_t.data_ptr(), stream)

    def step():
        local_step()
        exchange_step()

    def timed(fn, reps):
        ranks.fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ranks.fence()
        return ranks.max_over_ranks(time.perf_counter() - t0)

    for _ in range(warmup):
        step()
    # untimed side measurements: this rank's shard scan without the exchange, and the exchange (all-gather + merge) alone
    shard_alone_ms = timed(local_step, 3) / 3 * 1e3
    exchange_ms = timed(exchange_step, 10) / 10 * 1e3
    index.set_option("timing", 1)      # HIP events around the scan kernel, on the search stream, per step
    elapsed = timed(step, steps)
    st = index.stats()
    index.set_option("timing", 0)

    ms_per_step = elapsed / steps * 1e3
    qps_corpus = nq * steps / elapsed
    value = qps_corpus if strong else world * qps_corpus
    corpus_rows = n_total if strong else world * n_total
    transport = {"nccl": "RCCL all-gather (all_gather_into_tensor, device buffers)",
                 "none": "one rank: the all-gather is a copy"}.get(ranks.backend, f"{ranks.backend} all-gather staged through the host "
                                                                                 f"(shared-GPU rehearsal, NOT the product transport)")
    leg = {
        "metric": f"QPS ({workload}, k={k}" + (", strong scaling: whole-corpus queries/s)" if strong
                                              else ", weak scaling: query x shard scans/s)"),
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": scan_dtype_name(st), "data": "synthetic",
        "config": {"workload": f"{workload}: {n} rows x {d} dims on this GPU, {nq} queries, k={k}, {metric}; brute-force "
                               f"exact k-NN; inputs resident in HBM",
                   "rows_per_gpu": n, "corpus_rows": corpus_rows, "dim": d, "queries": nq, "k": k, "metric": metric,
                   "sharding": (f"row-sharded x{world} (strong scaling: the {n_total}-row corpus split over the ranks)"
                                if strong else
                                f"row-sharded x{world} (weak scaling: corpus = {world} x {n_total} rows)")
                               + f", {transport} of packed partial top-k + merge on every rank"},
        "rccl_ranks": int(ranks.dist.get_world_size()) if (world > 1 and ranks.backend == "nccl") else (1 if world == 1 else 0),
        "ranks": world, "collective_backend": ranks.backend,
        "roofline": roofline_of(st, nq, n, d, workload),
        "pipeline": pipeline_of(st, nq, build_s, float(n) * d * 4),
        "qps_whole_corpus": round(qps_corpus, 1),
        "shard_scan_alone_ms": round(shard_alone_ms, 4),
        "exchange_ms": round(exchange_ms, 4),
        "exchange_note": f"all-gather of {my_pack.numel() * 8} bytes per rank + merge of {world} partial lists "
                         f"per query, timed alone (10 repetitions, max over ranks)",
        "result_checksum": result_checksum(I_t),
    }
    if X is None:
        leg["recall@10_vs_float64_torch_sample"] = round(device_check(X_t, q_t, I_t, k, metric, lo, ranks=ranks), 6)
    else:      # host-generated shard: this rank's part of the merged result against the CPU oracle on its own rows is not the
        # merged truth; check instead that every returned id of a query sample is the exact neighbour among ALL shards
        leg["recall@10_vs_float64_torch_sample"] = round(
            device_check(torch.from_numpy(X).to(dev), q_t, I_t, k, metric, lo, ranks=ranks), 6)
    index.close()
    del X_t
    torch.cuda.empty_cache()
    return leg


def sharded_line(args, workload, vdbhip, torch, dev, rank, local_rank, world, stream):
    """N > 1 (or the one-rank rehearsal VDBHIP_BENCH_FORCE_SHARDED=1).  Headline = the SAME workload as the N = 1 line (sift1m,
    BASELINE.json configs[1]) under weak scaling: every rank holds its own 1M x 128 shard, `value` = N x queries / time -- so
    the driver's N = 1 / 2 / 4 / 8 series is one workload and its N = 1 point is the BENCH line.  The config-5 shard
    (12.5M x 768 per GPU, BASELINE.json configs[4]) rides along as also["marco12.5m"] with both its weak and its strong figure
    (N = 1 points: also["marco12.5m"] of the `--gpus 1` line)."""
    ranks = Ranks(torch, dev, world)
    strong = args.scaling == "strong"
    out = sharded_leg(args, workload, strong, vdbhip, torch, dev, rank, local_rank, ranks, stream, args.steps, args.warmup)
    out["scaling_reference"] = ("N = 1 point: the `--gpus 1` line's `value` for sift1m (the same shard, searched without the "
                                "gather + merge); for marco12.5m `also[\"marco12.5m\"]` of that line (value, result_checksum)")
    if workload == "sift1m" and not args.no_extras:
        also = {}
        for mode in ("weak", "strong"):
            also[mode] = sharded_leg(args, "marco12.5m", mode == "strong", vdbhip, torch, dev, rank, local_rank, ranks, stream,
                                     min(args.steps, 10), min(args.warmup, 2))
        out["also"] = {"marco12.5m": also}
    ranks.close()
    return out


if __name__ == "__main__":
    sys.exit(main())
