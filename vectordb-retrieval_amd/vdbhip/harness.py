"""Minimal experiment harness with the reference's call contract and result keys.

Mirrors the part of ExperimentRunner that drives the hot path (src/experiments/experiment_runner.py:259-488):
  build_index(train) timed -> batch_search(test[cursor:end], k) per query batch, wall-clock timed with
  time.time() (:431-433), batch = all queries when query_batch_size == 0 (:424), fallback to per-query
  search() on (AttributeError, NotImplementedError, TypeError, ValueError) (:442-455), indices normalised to
  (n,k) int64 with -1 padding (:381-418), qps = n_queries / total_query_time (:464), recall@{1,10,100} with
  k <= topk (evaluation.py:23-29, 47-50).
and the YAML shapes of configs/*.yaml (runner.py:95-155, 274-299): top-level `indexers`, `searchers`,
`algorithms`, `datasets[]`; an algorithm entry is {type, metric, ...kwargs} or {indexer_ref, searcher_ref}.
Datasets: the synthetic `random` recipe (dataset.py:473-504), or LOCAL files named in `dataset_options`
(`train_path` / `test_path` / optional `groundtruth_path`: `.fvecs` / `.ivecs` / `.npy`, read by io.py; missing ground
truth is computed through the HIP kernels) -- downloads are out of scope.
"""
from __future__ import annotations

import copy
import time
from typing import Any, Dict, List, Optional

import numpy as np

from . import datasets
from .metrics import latency_stats, recall_at_k
from .plugin_api import BaseAlgorithm, get_algorithm_instance


def resolve_modular_components(algorithms: Dict[str, Any], indexers: Dict[str, Any], searchers: Dict[str, Any]):
    """indexer_ref / searcher_ref -> inline dicts; default type 'Composite' (runner.py:274-299)."""
    out = {}
    for name, cfg in algorithms.items():
        cfg = copy.deepcopy(cfg)
        if "indexer_ref" in cfg or "searcher_ref" in cfg:
            iref, sref = cfg.pop("indexer_ref", None), cfg.pop("searcher_ref", None)
            if iref not in indexers:
                raise ValueError(f"Unknown indexer_ref '{iref}' for algorithm '{name}'")
            if sref not in searchers:
                raise ValueError(f"Unknown searcher_ref '{sref}' for algorithm '{name}'")
            cfg["indexer"] = copy.deepcopy(indexers[iref])
            cfg["searcher"] = copy.deepcopy(searchers[sref])
            cfg.setdefault("type", "Composite")
        out[name] = cfg
    return out


def normalize_batch_indices(batch_result: Any, expected_rows: int, expected_k: int) -> np.ndarray:
    """The caller-side return-shape contract of SURVEY 8 row a12 (the reference enforces it in
    experiment_runner.py:381-418), written from the contract:

      in   (distances, indices) pair -> its second member; a Python list of per-query id rows (ragged allowed,
           surplus rows ignored); or an array: (rows, k'), a flat (k',) answer to one query, or the degenerate
           (k, 1)-shaped column a single query sometimes comes back as
      out  (expected_rows, expected_k) int64, every row cut to expected_k ids and filled up with -1

    A result that cannot be read as one id row per query is a ValueError -- the exception the reference's harness
    turns into its per-query fallback."""
    ids = batch_result
    if isinstance(ids, tuple):
        if len(ids) != 2:
            raise ValueError("batch_search must return (distances, indices)")
        ids = ids[1]
    if isinstance(ids, list):
        rows = [np.ravel(np.asarray(r)) for r in ids[:expected_rows]]
    else:
        table = np.asarray(ids)
        if table.ndim == 1 or (table.ndim == 2 and expected_rows == 1 and table.shape[0] != 1
                               and table.shape[0] == expected_k):
            table = table.reshape(1, -1)
        if table.ndim != 2:
            raise ValueError("batch_search returned array with unexpected shape")
        if table.shape[0] != expected_rows:
            raise ValueError(f"batch_search returned {table.shape[0]} rows, expected {expected_rows}")
        if table.shape[1] == expected_k:                       # the common case: nothing to cut or fill
            return table.astype(np.int64, copy=False)
        rows = list(table)
    out = np.full((expected_rows, expected_k), -1, dtype=np.int64)
    for dst, src in zip(out, rows):
        keep = min(len(src), expected_k)
        dst[:keep] = src[:keep]
    return out


def run_single_algorithm(algorithm: BaseAlgorithm, train: np.ndarray, test: np.ndarray, ground_truth: np.ndarray,
                         topk: int, query_batch_size: int = 0, dataset: str = "random",
                         warmup_batches: int = 0) -> Dict[str, Any]:
    """`_run_single_algorithm` of the reference (experiment_runner.py:322-470).  `warmup_batches` > 0 adds the
    untimed warm-up the reference lacks (methodology/metrics_methodology.md:119-121): the
    first batch is replayed that many times before the timed pass, and the first-call time is reported beside it."""
    t0 = time.time()
    algorithm.build_index(train)
    build_time = time.time() - t0
    n = len(test)
    first_call_s = None
    if warmup_batches > 0 and n:
        wb = n if query_batch_size == 0 else min(query_batch_size, n)
        for w in range(warmup_batches):
            t0 = time.time()
            try:
                algorithm.batch_search(test[:wb], k=topk)
            except (AttributeError, NotImplementedError, TypeError, ValueError):
                break
            if w == 0:
                first_call_s = time.time() - t0
        counter = getattr(algorithm, "operation_counter", None)
        if isinstance(counter, dict):
            counter.clear()          # operations_per_query describes the timed pass only
    indices = np.full((n, topk), -1, dtype=np.int64)
    query_times = np.zeros(n)
    total = 0.0
    used_batch = False
    batch = n if query_batch_size == 0 else min(query_batch_size, n)
    if n:
        try:
            cur = 0
            while cur < n:
                end = min(cur + batch, n)
                t0 = time.time()
                res = algorithm.batch_search(test[cur:end], k=topk)
                dt = time.time() - t0
                indices[cur:end] = normalize_batch_indices(res, end - cur, topk)
                query_times[cur:end] = dt / max(end - cur, 1)
                total += dt
                cur = end
            used_batch = True
        except (AttributeError, NotImplementedError, TypeError, ValueError):
            indices.fill(-1)
            query_times.fill(0.0)
            total = 0.0
    if not used_batch:
        for i, q in enumerate(test):
            t0 = time.time()
            _, idx = algorithm.search(q, k=topk)
            dt = time.time() - t0
            query_times[i] = dt
            indices[i] = idx
            total += dt
    total = max(total, query_times.sum())
    mem = getattr(algorithm, "get_memory_usage", lambda: None)()
    metrics: Dict[str, Any] = {
        "algorithm": algorithm.get_name(), "parameters": algorithm.get_parameters(), "dataset": dataset,
        "n_train": int(train.shape[0]), "n_test": int(n), "dimensions": int(train.shape[1]), "topk": int(topk),
        "build_time_s": float(build_time), "total_query_time_s": float(total),
        "mean_query_time_ms": float(total / max(n, 1) * 1000.0), "qps": float(n / total) if total > 0 else 0.0,
        "index_memory_mb": float(mem) if mem else float(train.nbytes) / 2 ** 20, "used_batch_api": used_batch,
    }
    metrics["latency_s"] = latency_stats(query_times)       # keys of metrics.py:212-237 compute_cost_latency
    if first_call_s is not None:
        metrics["first_call_s"] = float(first_call_s)
        metrics["warmup_batches"] = int(warmup_batches)
    ops = algorithm.get_operations()
    if ops.get("ndis") and n:
        metrics["operations_per_query"] = ops["ndis"] / n     # picked up by evaluation.py:79-87 when present
    for k in (1, 10, 100):
        if k <= topk and ground_truth is not None:
            metrics[f"recall@{k}"] = float(recall_at_k(ground_truth, indices, k))
    if ground_truth is not None:
        metrics["recall"] = metrics.get(f"recall@{min(100, topk)}", max(
            (v for kk, v in metrics.items() if kk.startswith("recall@")), default=0.0))
    return {"metrics": metrics, "indices": indices}


def run_benchmark(config: Dict[str, Any]) -> Dict[str, Dict[str, Any]]:
    """Run every algorithm of a reference-shaped config on its `random` datasets; returns
    {dataset: {algorithm: metrics}}."""
    indexers, searchers = config.get("indexers", {}) or {}, config.get("searchers", {}) or {}
    base_algos = config.get("algorithms", {}) or {}
    results: Dict[str, Dict[str, Any]] = {}
    for entry in config.get("datasets", [{"name": "random"}]):
        name = entry.get("name", "random")
        opts = entry.get("dataset_options", {}) or {}
        gtk = int(opts.get("ground_truth_k", 100))
        if name == "random":
            dim, ntrain = int(opts.get("dimensions", 128)), int(opts.get("train_size", 10000))
            ntest = int(opts.get("test_size", 1000))
            train, test = datasets.random_reference(dim, ntrain, ntest, int(opts.get("seed", 42)))
            gt = np.stack([np.argsort(np.linalg.norm(train - q, axis=1))[:gtk] for q in test]).astype(np.int32)
        elif opts.get("train_path") and opts.get("test_path"):
            train, test, gt = load_local_dataset(opts, gtk, entry.get("metric") or config.get("metric") or "l2")
            dim, ntest = int(train.shape[1]), int(test.shape[0])
        else:
            raise ValueError(f"dataset '{name}' needs local files: give dataset_options.train_path / test_path "
                             f"(.fvecs / .npy); nothing is downloaded")
        topk = int(entry.get("topk", config.get("topk", 10)))
        nq = int(entry.get("n_queries", config.get("n_queries", ntest)))
        qbs = int(entry.get("query_batch_size", config.get("query_batch_size", 0)))
        if nq < ntest:   # experiment_runner.py:79, 138-153
            state = np.random.get_state()
            np.random.seed(int(config.get("seed", 42)))
            pick = np.random.choice(ntest, nq, replace=False)
            np.random.set_state(state)
            test, gt = test[pick], gt[pick]
        algos = copy.deepcopy(base_algos)
        for an, override in (entry.get("algorithms", {}) or {}).items():
            algos.setdefault(an, {}).update(override or {})
        metric = entry.get("metric")
        for cfg in algos.values():          # the dataset metric is forced onto every algorithm (runner.py:117-118)
            if metric:
                cfg["metric"] = metric
        algos = resolve_modular_components(algos, indexers, searchers)
        results[name] = {}
        for an, cfg in algos.items():
            cfg = copy.deepcopy(cfg)
            atype = cfg.pop("type")
            algo = get_algorithm_instance(atype, dim, name=an, **cfg)
            results[name][an] = run_single_algorithm(algo, train, test, gt, topk, qbs, name,
                                                     int(config.get("warmup_batches", 0)))["metrics"]
    return results


def _load_rows(path: str, limit: Optional[int], integer: bool = False) -> np.ndarray:
    from . import io

    p = str(path)
    if p.endswith(".fvecs"):
        return io.read_fvecs(p, limit)
    if p.endswith(".ivecs"):
        return io.read_ivecs(p, limit)
    if p.endswith(".npy"):
        return io.open_npy_rows(p, limit)
    raise ValueError(f"unsupported dataset file '{p}' (expected .fvecs, .ivecs or .npy)")


def load_local_dataset(opts: Dict[str, Any], gtk: int, metric: str):
    """(train, test, ground truth) from local files -- what Dataset.load does for sift1m / glove / msmarco once
    the files exist (dataset.py:376-471, 522-574, 1001-1052), with the .fvecs payload read as float32 bits.
    Ground truth comes from `groundtruth_path` (.ivecs / .npy) or, when absent or too narrow, from the exact
    search itself (`ground_truth`, the kernel path of dataset.py:858-964)."""
    train = _load_rows(opts["train_path"], opts.get("train_limit"))
    test = _load_rows(opts["test_path"], opts.get("test_limit"))
    if train.ndim != 2 or test.ndim != 2 or train.shape[1] != test.shape[1]:
        raise ValueError(f"train {train.shape} and test {test.shape} do not agree on the dimension")
    gt = None
    if opts.get("groundtruth_path") and not opts.get("train_limit"):
        gt = np.asarray(_load_rows(opts["groundtruth_path"], opts.get("test_limit")), dtype=np.int32)
        if gt.shape[0] != test.shape[0] or gt.shape[1] < min(gtk, 10):
            gt = None
    if gt is None:
        gt = ground_truth(train, test, k=min(gtk, train.shape[0]), metric=metric,
                          normalize=bool(opts.get("normalize_cosine_groundtruth", False)) and metric == "cosine")
    return train, test, gt


def ground_truth(train: np.ndarray, test: np.ndarray, k: int = 100, metric: str = "l2",
                 normalize: bool = False, device: int = 0) -> np.ndarray:
    """Brute-force ground truth through the HIP kernels: what the reference computes with
    per-query `argsort(norm(train - q))[:k]` (dataset.py:497-504, 657-663; int32 output) or with FAISS
    `IndexFlatIP/L2` for MS MARCO (dataset.py:858-964; `normalize=True` = normalize_cosine_groundtruth)."""
    from .algorithms import HipExactSearch, _safe_normalize

    x = np.ascontiguousarray(train, np.float32)
    q = np.ascontiguousarray(test, np.float32)
    if normalize:
        x, q = _safe_normalize(x), _safe_normalize(q)
    algo = HipExactSearch("ground_truth", x.shape[1], metric="l2" if metric == "l2" else "ip", device=device)
    algo.build_index(x)
    _, idx = algo.batch_search(q, k=k)
    return idx.astype(np.int32)


def load_config(path: str) -> Dict[str, Any]:
    import yaml

    with open(path) as f:
        return yaml.safe_load(f)
