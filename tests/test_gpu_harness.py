"""End-to-end plugin plumbing through a reference-shaped config (the idiom of the reference's
tests/test_benchmark_runner_modular.py:9-65): indexer_ref/searcher_ref resolution, result keys, counts."""
from __future__ import annotations

import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIG = {
    "seed": 42, "topk": 5, "n_queries": 5, "query_batch_size": 2,
    "indexers": {"hip_bf_l2": {"type": "HipBruteForceIndexer", "metric": "l2"},
                 "hip_ivf_l2": {"type": "HipIVFIndexer", "index_type": "IVF4,Flat", "metric": "l2", "nprobe": 4}},
    "searchers": {"hip_linear_l2": {"type": "HipLinearSearcher", "metric": "l2"},
                  "hip_ivf_search": {"type": "HipIVFSearcher", "metric": "l2"}},
    "algorithms": {"exact": {"indexer_ref": "hip_bf_l2", "searcher_ref": "hip_linear_l2", "metric": "l2"},
                   "exact_hip": {"type": "HipExactSearch", "metric": "l2"},
                   "ivf_flat": {"indexer_ref": "hip_ivf_l2", "searcher_ref": "hip_ivf_search", "metric": "l2"}},
    "datasets": [{"name": "random", "metric": "l2",
                  "dataset_options": {"dimensions": 3, "train_size": 32, "test_size": 5, "ground_truth_k": 5,
                                      "seed": 7}}],
}


def test_reference_shaped_config_end_to_end():
    from vdbhip import harness

    res = harness.run_benchmark(CONFIG)["random"]
    assert set(res) == {"exact", "exact_hip", "ivf_flat"}
    for name, m in res.items():
        for key in ("qps", "mean_query_time_ms", "total_query_time_s", "build_time_s", "index_memory_mb", "recall@1",
                    "n_train", "n_test", "dimensions", "topk", "parameters"):
            assert key in m, (name, key)
        assert m["n_train"] == 32 and m["n_test"] == 5 and m["topk"] == 5 and m["used_batch_api"]
        assert m["recall@1"] == 1.0 and m["recall"] == 1.0      # exact; IVF probes all 4 lists
        json.dumps(m)
    assert res["exact"]["parameters"]["indexer"]["type"] == "HipBruteForceIndexer"
    assert res["exact_hip"]["operations_per_query"] == 32


def test_harness_swallows_valueerror_like_the_reference():
    """A ValueError from batch_search silently degrades to per-query search (experiment_runner.py:442-455):
    the reason the HIP plugins only ever raise RuntimeError at search time."""
    from vdbhip import BaseAlgorithm, harness

    class Flaky(BaseAlgorithm):
        def build_index(self, vectors, metadata=None):
            self.v = vectors
            self.index_built = True

        def batch_search(self, queries, k=10):
            raise ValueError("boom")

        def search(self, query, k=10):
            d = np.linalg.norm(self.v - query, axis=1)
            i = np.argsort(d)[:k]
            return d[i], i

    rng = np.random.default_rng(0)
    X, Q = rng.standard_normal((20, 3)).astype(np.float32), rng.standard_normal((4, 3)).astype(np.float32)
    gt = np.stack([np.argsort(np.linalg.norm(X - q, axis=1))[:3] for q in Q])
    out = harness.run_single_algorithm(Flaky("f", 3), X, Q, gt, 3)
    assert out["metrics"]["used_batch_api"] is False and out["metrics"]["recall@1"] == 1.0


def test_ground_truth_through_the_kernel(oracle, tmp_path):
    """dataset.py:497-504: GT = argsort(norm(train - q))[:k], int32 -- here via the HIP path, k = 100/200;
    the corpus is uploaded straight from a read-only .npy memory map (dataset.py:397, 414)."""
    from vdbhip import harness, io
    from oracle import ref_semantics as rs

    X, Q = rs.random_dataset(32, 50000, 300, 11)
    np.save(tmp_path / "train.npy", X)
    mm = io.open_npy_rows(tmp_path / "train.npy")
    assert not mm.flags.writeable
    gt = harness.ground_truth(mm, Q, k=200, metric="l2")
    assert gt.dtype == np.int32 and gt.shape == (300, 200)
    np.testing.assert_array_equal(gt, oracle.knn(X, Q, 200, "l2")[1].astype(np.int32))
    np.testing.assert_array_equal(gt[:20, :100], rs.ground_truth_l2(X, Q[:20], 100))
    gtc = harness.ground_truth(X, Q, k=100, metric="ip", normalize=True)
    np.testing.assert_array_equal(gtc, oracle.knn(rs.safe_normalize(X), rs.safe_normalize(Q), 100, "ip")[1].astype(np.int32))


def test_local_file_datasets_fvecs_and_npy(oracle, tmp_path):
    """Datasets from local .fvecs / .ivecs / .npy files through the reference-shaped config:
    the stored ground truth is used when present, otherwise it is computed by the exact kernel path."""
    from vdbhip import harness, io

    rng = np.random.default_rng(4)
    X = np.rint(rng.standard_normal((40000, 24)) * 20).astype(np.float32)
    Q = np.rint(rng.standard_normal((120, 24)) * 20).astype(np.float32)
    gt = oracle.knn(X, Q, 100, "l2")[1].astype(np.int32)
    io.write_fvecs(tmp_path / "base.fvecs", X)
    io.write_fvecs(tmp_path / "query.fvecs", Q)
    io.write_ivecs(tmp_path / "gt.ivecs", gt)
    np.save(tmp_path / "base.npy", X)
    np.save(tmp_path / "query.npy", Q)
    base = {"topk": 10, "algorithms": {"exact": {"type": "HipExactSearch", "metric": "l2"},
                                        "ivf": {"type": "HipApproximateSearch", "index_type": "IVF64,Flat",
                                                "metric": "l2", "nprobe": 64}}}
    cfg1 = dict(base, datasets=[{"name": "sift_local", "metric": "l2", "dataset_options": {
        "train_path": str(tmp_path / "base.fvecs"), "test_path": str(tmp_path / "query.fvecs"),
        "groundtruth_path": str(tmp_path / "gt.ivecs")}}])
    cfg2 = dict(base, datasets=[{"name": "npy_local", "metric": "l2", "dataset_options": {
        "train_path": str(tmp_path / "base.npy"), "test_path": str(tmp_path / "query.npy"), "ground_truth_k": 50}}])
    for cfg, name in ((cfg1, "sift_local"), (cfg2, "npy_local")):
        res = harness.run_benchmark(cfg)[name]
        assert res["exact"]["recall@10"] == 1.0 and res["exact"]["n_train"] == 40000 and res["exact"]["n_test"] == 120
        assert res["ivf"]["recall@10"] == 1.0          # nprobe = nlist: IVF degenerates to brute force
    with pytest.raises(ValueError, match="needs local files"):
        harness.run_benchmark(dict(base, datasets=[{"name": "sift1m"}]))


def test_row_block_ingestion_from_a_memmap_larger_than_the_staging_buffer(vdb, oracle, tmp_path):
    """vdb_add / vdb_ivf_add stream the host rows in blocks through two pinned staging buffers (dataset.py:376-471 keeps
    its corpora as np.memmap): with a 1 MiB block a 24 MB memory-mapped corpus goes up in 24+ blocks, and the index is
    the same as the one built from the in-memory array."""
    rng = np.random.default_rng(8)
    X = rng.standard_normal((61_237, 100)).astype(np.float32)      # D = 100: rows are re-pitched to 100 floats on the device
    Q = rng.standard_normal((300, 100)).astype(np.float32)
    path = tmp_path / "corpus.npy"
    np.save(path, X)
    mm = np.load(path, mmap_mode="r")
    idx = vdb.FlatIndex(100, "l2", 0)
    idx.set_option("upload_block_mb", 1)
    idx.add(mm, id_base=11)
    rows_per_block = (1 << 20) // (100 * 4)
    assert idx.stats()["upload_blocks"] == -(-len(X) // rows_per_block) == 24
    D, I = idx.search(Q, 10)
    Do, Io = oracle.knn(X, Q, 10, "l2", id_base=11)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    idx.close()
    C = X[rng.choice(len(X), 64, replace=False)].copy()
    ivf = vdb.IVFFlatIndex(100, 64, "l2", 0)
    ivf.set_centroids(C)
    ivf.set_option("upload_block_mb", 1)
    ivf.add(mm)
    assert ivf.stats()["upload_blocks"] > 20
    np.testing.assert_array_equal(ivf.assignment(), oracle.ivf_assign(C, X, "l2"))
    ivf.set_nprobe(8)
    D, I = ivf.search(Q, 10)
    Do, Io = oracle.ivf_search(X, C, ivf.assignment(), Q, 10, 8, "l2")
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    ivf.close()


def test_build_index_reserves_the_workspace_of_the_first_batch(vdb, oracle):
    """The reference times its very first batch_search (experiment_runner.py:431-437): `build_index` sizes the search
    workspace for `reserve_queries` queries (default 10 000) so that the first call allocates nothing; results as ever."""
    rng = np.random.default_rng(12)
    X = rng.standard_normal((60_000, 64)).astype(np.float32)
    Q = rng.standard_normal((3000, 64)).astype(np.float32)
    algo = vdb.get_algorithm_instance("HipExactSearch", 64, name="r", metric="l2")
    algo.build_index(X)
    before = algo.index.stats()["bytes_resident"]
    D, I = algo.batch_search(Q, k=10)
    assert algo.index.stats()["bytes_resident"] == before         # nothing grew during the first search
    Do, Io = oracle.knn(X, Q[:64], 10, "l2")
    np.testing.assert_array_equal(I[:64], Io)
    lazy = vdb.get_algorithm_instance("HipExactSearch", 64, name="r0", metric="l2", reserve_queries=0)
    lazy.build_index(X)
    small = lazy.index.stats()["bytes_resident"]
    assert small < before
    D2, I2 = lazy.batch_search(Q, k=10)
    np.testing.assert_array_equal(I2, I)
    np.testing.assert_array_equal(D2, D)
    assert lazy.index.stats()["bytes_resident"] > small
    ivf = vdb.get_algorithm_instance("HipApproximateSearch", 64, name="a", index_type="IVF64,Flat", metric="l2", nprobe=8)
    ivf.build_index(X)
    b0 = ivf.index.stats()["bytes_resident"]
    ivf.batch_search(Q, k=10)
    assert ivf.index.stats()["bytes_resident"] == b0
