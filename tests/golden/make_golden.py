#!/usr/bin/env python3
"""Generate golden input/output vectors from the REFERENCE implementation.

Runs only in the build container (needs /root/reference); the fixtures it
writes (tests/golden/*.npz, *.json) are DATA: seeds/inputs and the outputs the
reference's own NumPy hot path produced for them.  No reference source text is
copied.

Reference entry points exercised (file:line relative to /root/reference):
  * src/algorithms/modular.py:121-133   BruteForceIndexer.build
  * src/algorithms/modular.py:312-390   LinearSearcher.attach/search/batch_search
  * src/algorithms/modular.py:554-622   CompositeAlgorithm
  * src/algorithms/modular.py:109-118   _safe_normalize / _ensure_float32
  * src/benchmark/dataset.py:473-504    Dataset._generate_random_dataset
  * src/benchmark/metrics.py:4-34       recall_at_k

`src/algorithms/__init__.py` eagerly imports faiss (absent here), so the
package object is stubbed and only the faiss-free module `modular` is imported
(the recipe of SURVEY.md section 8c).  ExactSearch / FaissSearcher cannot run
here (no faiss): their conventions are pinned by the reference's source and
FAISS's documented semantics only (see oracle/README.md, "parity pins").

Usage:  PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py
"""
from __future__ import annotations

import hashlib
import importlib
import json
import os
import sys
import types
from pathlib import Path

import numpy as np

REF = Path(os.environ.get("VDB_REFERENCE", "/root/reference"))
OUT = Path(__file__).resolve().parent


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    import src  # noqa: F401  (namespace of the reference)

    pkg = types.ModuleType("src.algorithms")
    pkg.__path__ = [str(REF / "src" / "algorithms")]
    sys.modules["src.algorithms"] = pkg
    modular = importlib.import_module("src.algorithms.modular")
    metrics = importlib.import_module("src.benchmark.metrics")
    dataset = importlib.import_module("src.benchmark.dataset")
    return modular, metrics, dataset


def _sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _composite(modular, dim, metric):
    return modular.CompositeAlgorithm(
        name=f"bf_{metric}",
        dimension=dim,
        metric=metric,
        indexer={"type": "BruteForceIndexer", "metric": metric},
        searcher={"type": "LinearSearcher", "metric": metric},
    )


def _run(modular, X, Q, k, metric, batch=None):
    algo = _composite(modular, X.shape[1], metric)
    algo.build_index(X)
    if batch is None:
        return algo.batch_search(Q, k=k)
    Ds, Is = [], []
    for s in range(0, len(Q), batch):
        d, i = algo.batch_search(Q[s:s + batch], k=k)
        Ds.append(d)
        Is.append(i)
    return np.concatenate(Ds), np.concatenate(Is)


def main() -> None:
    modular, metrics, dataset = _import_reference()
    manifest = {"generator": "tests/golden/make_golden.py", "numpy": np.__version__, "cases": {}}

    # ---- KAT-0: the reference's own known-answer test (tests/test_composite_algorithm.py:29-58)
    X0 = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], dtype=np.float32)
    Q0 = np.array([[0.1, 0.1], [0.9, 0.2]], dtype=np.float32)
    out = {"X": X0, "Q": Q0}
    for m in ("l2", "cosine", "ip"):
        d, i = _run(modular, X0, Q0, 2, m)
        out[f"D_{m}"], out[f"I_{m}"] = d, i
    np.savez(OUT / "kat0_ref_unit.npz", **out)
    manifest["cases"]["kat0_ref_unit"] = {"k": 2, "note": "reference unit test corpus 4x2"}

    # ---- KAT-1: RandomState(0) 1000x16, 5 queries, k=3
    rs = np.random.RandomState(0)
    X1 = rs.randn(1000, 16).astype(np.float32)
    Q1 = rs.randn(5, 16).astype(np.float32)
    out = {}
    for m in ("l2", "cosine", "ip"):
        d, i = _run(modular, X1, Q1, 3, m)
        out[f"D_{m}"], out[f"I_{m}"] = d, i
    np.savez(OUT / "kat1_rs0_1000x16.npz", **out)
    manifest["cases"]["kat1_rs0_1000x16"] = {
        "recipe": "rs=RandomState(0); X=rs.randn(1000,16).f32; Q=rs.randn(5,16).f32", "k": 3,
        "sha_X": _sha(X1), "sha_Q": _sha(Q1)}

    # ---- KAT-2: BASELINE config 1 shape through Dataset._generate_random_dataset
    ds = dataset.Dataset("random", data_dir="/tmp/vdb_golden_data",
                         options={"dimensions": 128, "train_size": 10000, "test_size": 100,
                                  "ground_truth_k": 100, "seed": 42})
    ds._generate_random_dataset()
    X2, Q2, GT2 = ds.train_vectors, ds.test_vectors, ds.ground_truth
    out = {"GT": GT2}
    rec = {}
    for m in ("l2", "cosine", "ip"):
        d, i = _run(modular, X2, Q2, 10, m, batch=20)
        out[f"D_{m}"], out[f"I_{m}"] = d, i
        rec[m] = float(metrics.recall_at_k(GT2, i, 10))
    np.savez(OUT / "kat2_random_10000x128.npz", **out)
    manifest["cases"]["kat2_random_10000x128"] = {
        "recipe": "np.random.seed(42); X=randn(10000,128).f32; Q=randn(100,128).f32 (dataset.py:491-495)",
        "k": 10, "sha_X": _sha(X2), "sha_Q": _sha(Q2), "recall_at_10_vs_GT": rec}

    # ---- KAT-3: smoke-config shape (configs/benchmark_config_smoke.yaml:87-95), large k
    ds = dataset.Dataset("random", data_dir="/tmp/vdb_golden_data",
                         options={"dimensions": 64, "train_size": 20000, "test_size": 256,
                                  "ground_truth_k": 100, "seed": 7})
    ds._generate_random_dataset()
    X3, Q3, GT3 = ds.train_vectors, ds.test_vectors, ds.ground_truth
    d, i = _run(modular, X3, Q3, 100, "l2", batch=32)
    np.savez_compressed(OUT / "kat3_smoke_20000x64_k100.npz", D_l2=d, I_l2=i, GT=GT3)
    manifest["cases"]["kat3_smoke_20000x64_k100"] = {
        "recipe": "np.random.seed(7); X=randn(20000,64).f32; Q=randn(256,64).f32", "k": 100,
        "sha_X": _sha(X3), "sha_Q": _sha(Q3),
        "recall_at_100_vs_GT": float(metrics.recall_at_k(GT3, i, 100))}

    # ---- EDGE: ties, duplicates, zero-norm rows, k > N, float64 / Fortran inputs
    Xe = np.array([[0, 0], [1, 0], [0, 1], [1, 1], [0, 0]], dtype=np.float32)
    Qe = np.array([[0.1, 0.1], [0, 0]], dtype=np.float32)
    out = {"X": Xe, "Q": Qe}
    for m in ("l2", "cosine", "ip"):
        d, i = _run(modular, Xe, Qe, 7, m)
        out[f"D_{m}_k7"], out[f"I_{m}_k7"] = d, i
        d, i = _run(modular, Xe, Qe, 3, m)
        out[f"D_{m}_k3"], out[f"I_{m}_k3"] = d, i
    # single-query search() returns 1-D arrays
    algo = _composite(modular, 2, "l2")
    algo.build_index(Xe)
    d1, i1 = algo.search(Qe[0], k=3)
    out["D_search1d"], out["I_search1d"] = d1, i1
    # float64 + Fortran-ordered inputs are accepted
    X64 = np.asfortranarray(X1[:50].astype(np.float64))
    d, i = _run(modular, X64, Q1.astype(np.float64), 4, "l2")
    out["D_f64F"], out["I_f64F"] = d, i
    np.savez(OUT / "edge_cases.npz", **out)
    manifest["cases"]["edge_cases"] = {"note": "ties/duplicates/zero-norm/k>N/1-D search/f64 Fortran"}

    # ---- error conventions of the boundary (modular.py:316-320, 571-572, 615-621, 387)
    errs = {}
    a = _composite(modular, 2, "l2")
    try:
        a.batch_search(Qe, 2)
    except Exception as e:  # noqa: BLE001
        errs["search_before_build"] = [type(e).__name__, str(e)]
    try:
        modular.CompositeAlgorithm(name="x", dimension=2, indexer={}, searcher={"type": "LinearSearcher"})
    except Exception as e:  # noqa: BLE001
        errs["empty_indexer"] = [type(e).__name__, str(e)]
    try:
        a3 = _composite(modular, 3, "l2")
        a3.build_index(Xe)
    except Exception as e:  # noqa: BLE001
        errs["dim_mismatch"] = [type(e).__name__, str(e)]
    try:
        ad = _composite(modular, 2, "dot")
        ad.build_index(Xe)
        ad.batch_search(Qe, 2)
    except Exception as e:  # noqa: BLE001
        errs["bad_metric"] = [type(e).__name__, str(e)]
    try:
        modular.get_searcher_class("Nope")
    except Exception as e:  # noqa: BLE001
        errs["unknown_searcher"] = [type(e).__name__, str(e)[:40]]
    manifest["errors"] = errs

    # recall_at_k known answers (metrics.py:4-34)
    gt = np.array([[1, 2, 3, 4], [5, 6, 7, 8]])
    pr = np.array([[1, 9, 3, 0], [8, 7, 6, 5]])
    manifest["recall_at_k"] = {
        "gt": gt.tolist(), "pred": pr.tolist(),
        "r1": float(metrics.recall_at_k(gt, pr, 1)), "r2": float(metrics.recall_at_k(gt, pr, 2)),
        "r4": float(metrics.recall_at_k(gt, pr, 4)), "r10": float(metrics.recall_at_k(gt, pr, 10))}

    # published result points of the reference (committed result JSONs + the config of that run): the only pins the
    # reference holds for its FAISS-backed IVF path
    import yaml

    run = REF / "benchmark_results" / "benchmark_20260305_070532" / "random"
    cfg = yaml.safe_load((run / "random_config.yaml").read_text())
    ivf = json.loads((run / "ivf_flat_results.json").read_text())
    exact = json.loads((run / "exact_results.json").read_text())
    ip = ivf["parameters"]["indexer"]["params"]
    manifest["published_points"] = {"random_ivf_flat": {
        "source": "benchmark_results/benchmark_20260305_070532/random/{ivf_flat,exact}_results.json + random_config.yaml",
        "dataset_options": cfg["dataset_options"], "config_seed": cfg["seed"], "n_queries": cfg["n_queries"],
        "topk": cfg["topk"], "index_type": ip["index_type"], "nprobe": ip["nprobe"],
        "ivf_flat": {"recall@1": ivf["recall@1"], "recall@10": ivf["recall@10"]},
        "exact": {"recall@1": exact["recall@1"], "recall@10": exact["recall@10"]}}}

    (OUT / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
    print(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
