"""ONE index over several devices inside one process (vdb_create_multi, csrc/multi.inc) -- the form the reference's
single-process harness can reach (experiment_runner.py:329-331, 428-434; algorithms/__init__.py:37-47).

The test box has one GPU, so the shards share it (`device_ids=[0, 0, 0]` is legal: a device may be listed more than once).
What is pinned: results bit-identical to the single-device index AND to the oracle, for flat L2 / IP (int8, fp16, K-loop and
exact-kernel paths), IVF-Flat, uneven and empty shards, appends, the device-pointer entry points, the plugins and the harness.
"""
from __future__ import annotations

import ctypes
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vdb():
    import vdbhip

    return vdbhip


def _gauss(n, d, nq, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, d)).astype(np.float32), rng.standard_normal((nq, d)).astype(np.float32)


def _sift_like(n, d, nq, seed):
    rng = np.random.default_rng(seed)
    f = lambda m: np.clip(np.round(rng.gamma(0.6, 40.0, size=(m, d))), 0, 218).astype(np.float32)
    return f(n), f(nq)


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("shape", [
    ("gauss", 100_003, 128, 700, 10, [0, 0, 0]),        # fp16 scan in every shard, uneven blocks (33 335 / 33 335 / 33 333)
    ("sift", 150_000, 128, 1000, 10, [0, 0]),           # int8 scan
    ("gauss", 70_000, 384, 300, 20, [0, 0, 0]),         # K-loop scan (< 32 768 rows per shard at ndev 3: the exact kernels)
    ("gauss", 3_001, 32, 50, 5, [0, 0, 0, 0]),          # tiny shards: exhaustive exact kernel
    ("gauss", 5, 16, 7, 8, [0, 0, 0]),                  # k > N, and shards without rows (5 rows over 3: 2 / 2 / 1)
    ("gauss", 2, 16, 3, 4, [0, 0, 0]),                  # an EMPTY shard (2 rows over 3: 1 / 1 / 0)
])
def test_flat_multi_equals_single_and_oracle(vdb, oracle, metric, shape):
    kind, n, d, nq, k, devs = shape
    X, Q = (_sift_like if kind == "sift" else _gauss)(n, d, nq, seed=n + d)
    multi = vdb.FlatIndex(d, metric, devs)
    multi.add(X, id_base=77)
    assert multi.ntotal == n and multi.stats()["ndevices"] == len(devs)
    D, I = multi.search(Q, k)
    single = vdb.FlatIndex(d, metric, 0)
    single.add(X, id_base=77)
    Ds, Is = single.search(Q, k)
    np.testing.assert_array_equal(I, Is)
    np.testing.assert_array_equal(D, Ds)
    Do, Io = oracle.knn(X, Q, k, metric, id_base=77)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    st = multi.stats()
    assert st["ntotal"] == n and st["bytes_resident"] > 0
    if kind == "sift":
        assert st["has_i8_copy"] == 1 and st["scan_dtype"] == 1 and st["last_path_name"] == "mfma_scan"
    # the remote-shard code path (own query copy, own packed buffer, peer copy into the gather slot) on this one-GPU box
    multi.set_option("multi_stage_all", 1)
    D2, I2 = multi.search(Q, k)
    np.testing.assert_array_equal(I2, Io)
    np.testing.assert_array_equal(D2, Do)
    # a single query and a second batch through the same buffers
    D1, I1 = multi.search(Q[:1], k)
    np.testing.assert_array_equal(I1, Io[:1])
    np.testing.assert_array_equal(D1, Do[:1])
    multi.close()
    single.close()


def test_appends_keep_global_ids_in_insertion_order(vdb, oracle):
    """vdb_add appends on a multi-device handle too: every add is cut into blocks of its own, so a shard holds several id
    ranges -- the merge kernel's segment table turns local rows into `id_base + insertion order`."""
    X, Q = _gauss(90_011, 64, 257, 5)
    cuts = [0, 40_000, 40_003, 70_001, 90_011]
    multi = vdb.FlatIndex(64, "l2", [0, 0, 0])
    for a, b in zip(cuts[:-1], cuts[1:]):
        multi.add(X[a:b], id_base=1000)
        assert multi.ntotal == b
        D, I = multi.search(Q, 10)
        Do, Io = oracle.knn(X[:b], Q, 10, "l2", id_base=1000)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
    with pytest.raises(ValueError, match="id base"):
        multi.add(X[:10], id_base=5)
    multi.reset()
    assert multi.ntotal == 0
    with pytest.raises(RuntimeError, match="not been built"):
        multi.search(Q, 10)
    multi.add(X[:5000], id_base=0)           # after reset another id base is fine
    np.testing.assert_array_equal(multi.search(Q, 3)[1], oracle.knn(X[:5000], Q, 3, "l2")[1])
    multi.close()


def test_device_pointer_entry_points_and_partials(vdb, oracle):
    """vdb_search_device / vdb_search_partial_device / vdb_add_device on a multi-device handle: device pointers are memory
    of the primary device; a partial of the whole multi index merges with other partials like any shard's."""
    import torch

    from vdbhip.index import merge_packed_partials_device

    dev = torch.device("cuda", 0)
    X, Q = _gauss(120_000, 128, 600, 9)
    Xd, Qd = torch.from_numpy(X).to(dev), torch.from_numpy(Q).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    k = 10
    half = 60_000
    a = vdb.FlatIndex(128, "ip", [0, 0, 0])
    a.add_device(Xd[:half].data_ptr(), half, id_base=0, stream=st)
    b = vdb.FlatIndex(128, "ip", [0, 0])
    b.add_device(Xd[half:].data_ptr(), len(X) - half, id_base=half, stream=st)
    Dd = torch.empty((len(Q), k), dtype=torch.float32, device=dev)
    Id = torch.empty((len(Q), k), dtype=torch.int64, device=dev)
    a.search_device(Qd.data_ptr(), len(Q), k, Dd.data_ptr(), Id.data_ptr(), st)
    torch.cuda.synchronize()
    Do, Io = oracle.knn(X[:half], Q, k, "ip")
    np.testing.assert_array_equal(Id.cpu().numpy(), Io)
    np.testing.assert_array_equal(Dd.cpu().numpy(), Do)
    # partials of the two multi indices -> one packed buffer -> the flat merge
    pack = torch.empty((2, 2, len(Q), k), dtype=torch.int64, device=dev)
    for j, idx in enumerate((a, b)):
        idx.search_partial_device(Qd.data_ptr(), len(Q), k, pack[j, 0].data_ptr(), pack[j, 1].data_ptr(), st)
    merge_packed_partials_device("ip", 0, pack.data_ptr(), 2, len(Q), k, Dd.data_ptr(), Id.data_ptr(), st)
    torch.cuda.synchronize()
    Do, Io = oracle.knn(X, Q, k, "ip")
    np.testing.assert_array_equal(Id.cpu().numpy(), Io)
    np.testing.assert_array_equal(Dd.cpu().numpy(), Do)
    # a serving loop through the same buffers: queries rewritten in place (second half: every shard on the remote-shard path)
    for r in range(4):
        if r == 2:
            a.set_option("multi_stage_all", 1)
        Qd.copy_(torch.from_numpy(np.roll(Q, r + 1, axis=0)).to(dev))
        a.search_device(Qd.data_ptr(), len(Q), k, Dd.data_ptr(), Id.data_ptr(), st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(Id.cpu().numpy(), np.roll(oracle.knn(X[:half], Q, k, "ip")[1], r + 1, axis=0))
    a.close()
    b.close()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_ivf_multi_equals_single_and_oracle(vdb, oracle, metric):
    rng = np.random.default_rng(3)
    n, d, nlist, nq, k = 60_001, 64, 128, 333, 10
    centers = rng.standard_normal((40, d)).astype(np.float32) * 3
    X = (centers[rng.integers(0, 40, n)] + rng.standard_normal((n, d))).astype(np.float32)
    Q = (centers[rng.integers(0, 40, nq)] + rng.standard_normal((nq, d))).astype(np.float32)
    C = X[rng.choice(n, nlist, replace=False)].copy()
    multi = vdb.IVFFlatIndex(d, nlist, metric, [0, 0, 0])
    multi.set_centroids(C)
    np.testing.assert_array_equal(multi.centroids(), C)
    multi.add(X[:25_000], id_base=500)
    multi.add(X[25_000:], id_base=500)                  # append: lists rebuilt inside every shard
    assert multi.ntotal == n
    lor = multi.assignment()
    np.testing.assert_array_equal(lor, oracle.ivf_assign(C, X, metric))
    for nprobe in (1, 8, nlist):
        multi.set_nprobe(nprobe)
        D, I = multi.search(Q, k)
        Do, Io = oracle.ivf_search(X, C, lor, Q, k, nprobe, metric, id_base=500)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
    st = multi.stats()
    assert st["last_path_name"] == "ivf" and st["nlist"] == nlist and st["ndevices"] == 3
    # stored assignment (what load_index does) + the library's own k-means, trained once and shared by the shards
    again = vdb.IVFFlatIndex(d, nlist, metric, [0, 0])
    again.set_centroids(C)
    again.add(X, id_base=500, list_of_row=lor)
    again.set_nprobe(8)
    multi.set_nprobe(8)
    np.testing.assert_array_equal(again.search(Q, k)[1], multi.search(Q, k)[1])
    again.close()
    multi.close()
    trained = vdb.IVFFlatIndex(d, 64, metric, [0, 0, 0])
    trained.train(X, niter=5, seed=7)
    ref = vdb.IVFFlatIndex(d, 64, metric, 0)
    ref.train(X, niter=5, seed=7)
    np.testing.assert_array_equal(trained.centroids(), ref.centroids())
    trained.add(X)
    ref.add(X)
    trained.set_nprobe(4)
    ref.set_nprobe(4)
    Dm, Im = trained.search(Q, k)
    Dr, Ir = ref.search(Q, k)
    np.testing.assert_array_equal(Im, Ir)
    np.testing.assert_array_equal(Dm, Dr)
    trained.close()
    ref.close()


def test_what_a_multi_handle_refuses(vdb):
    X, Q = _gauss(4000, 16, 4, 1)
    m = vdb.FlatIndex(16, "l2", [0, 0])
    m.add(X)
    with pytest.raises(RuntimeError, match="multi-device"):
        m.set_option("graph", 1)
    m.set_option("force_path", 1)                        # forwarded to every shard
    m.search(Q, 3)
    assert m.stats()["last_path_name"] == "exact_scan"
    with pytest.raises(ValueError, match="unknown option"):
        m.set_option("no_such_option", 1)
    m.close()
    with pytest.raises(ValueError, match="no such GPU"):
        vdb.FlatIndex(16, "l2", [0, 99])
    with pytest.raises(ValueError):
        vdb.FlatIndex(16, "l2", [])


def test_plugins_honour_device_ids(vdb, oracle):
    """`device_ids` of length > 1 is no longer cut to its first entry."""
    from oracle import ref_semantics as rs

    X, Q = _gauss(50_000, 48, 120, 12)
    algo = vdb.get_algorithm_instance("HipExactSearch", 48, name="exact8", metric="l2", device_ids=[0, 0, 0])
    algo.build_index(X)
    assert algo.index.stats()["ndevices"] == 3 and algo.get_parameters()["device_ids"] == [0, 0, 0]
    D, I = algo.batch_search(Q, 10)
    Do, Io = oracle.knn(X, Q, 10, "l2")
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    d1, i1 = algo.search(Q[0], 10)
    np.testing.assert_array_equal(i1, Io[0])
    assert algo.get_memory_usage() > 0
    json.dumps(algo.get_parameters())
    # Composite: BruteForceIndexer + LinearSearcher conventions over three shards (cosine: normalise + IP, negated)
    comp = vdb.get_algorithm_instance("Composite", 48, name="exact", metric="cosine",
                                      indexer={"type": "HipBruteForceIndexer", "metric": "cosine", "device_ids": [0, 0, 0]},
                                      searcher={"type": "HipLinearSearcher", "metric": "cosine"})
    comp.build_index(X)
    Dc, Ic = comp.batch_search(Q, 10)
    Dr, Ir = oracle.knn(rs.safe_normalize(X), rs.safe_normalize(Q), 10, "ip")     # canonical order of the same operands
    np.testing.assert_array_equal(Ic, Ir)
    np.testing.assert_array_equal(Dc, -Dr)
    # IVF through ApproximateSearch
    ivf = vdb.get_algorithm_instance("HipApproximateSearch", 48, name="ivf", index_type="IVF32,Flat", metric="l2",
                                     nprobe=32, device_ids=[0, 0], niter=4)
    ivf.build_index(X)
    assert ivf.index.stats()["ndevices"] == 2
    np.testing.assert_array_equal(ivf.batch_search(Q, 10)[1], Io)      # every list probed == brute force


def test_harness_drives_a_sharded_yaml_entry():
    """A reference-shaped config whose algorithm entry asks for two devices runs through harness.run_benchmark unchanged
    (what scripts/run_full_benchmark.py does with a YAML file: runner.py:177-181)."""
    from vdbhip import harness

    cfg = {
        "seed": 42, "topk": 10, "n_queries": 64, "query_batch_size": 0,
        "indexers": {"hip_ivf": {"type": "HipIVFIndexer", "index_type": "IVF16,Flat", "metric": "l2", "nprobe": 16,
                                 "device_ids": [0, 0, 0]}},
        "searchers": {"hip_ivf_search": {"type": "HipIVFSearcher", "metric": "l2"}},
        "algorithms": {"exact_2gpu": {"type": "HipExactSearch", "metric": "l2", "device_ids": [0, 0]},
                       "exact_1gpu": {"type": "HipExactSearch", "metric": "l2"},
                       "ivf_3gpu": {"indexer_ref": "hip_ivf", "searcher_ref": "hip_ivf_search", "metric": "l2"}},
        "datasets": [{"name": "random", "metric": "l2",
                      "dataset_options": {"dimensions": 32, "train_size": 40_000, "test_size": 64, "ground_truth_k": 10,
                                          "seed": 7}}],
    }
    res = harness.run_benchmark(cfg)["random"]
    assert set(res) == {"exact_2gpu", "exact_1gpu", "ivf_3gpu"}
    for name, m in res.items():
        assert m["recall@10"] == 1.0 and m["used_batch_api"] and m["n_train"] == 40_000, (name, m)
        json.dumps(m)
    assert res["exact_2gpu"]["parameters"]["device_ids"] == [0, 0]
    assert res["exact_2gpu"]["index_memory_mb"] > 0


def test_ivf_shards_without_rows(vdb, oracle):
    """Fewer rows than shards: blocks of 2 / 2 / 1 / 0 rows -- an IVF shard with nothing filed answers with padding only."""
    rng = np.random.default_rng(4)
    X = rng.standard_normal((5, 8)).astype(np.float32)
    Q = rng.standard_normal((3, 8)).astype(np.float32)
    C = X[:2].copy()
    ix = vdb.IVFFlatIndex(8, 2, "l2", [0, 0, 0, 0])
    ix.set_centroids(C)
    ix.add(X, id_base=100)
    lor = ix.assignment()
    np.testing.assert_array_equal(lor, oracle.ivf_assign(C, X, "l2"))
    for nprobe, k in ((2, 7), (1, 2)):
        ix.set_nprobe(nprobe)
        D, I = ix.search(Q, k)
        Do, Io = oracle.ivf_search(X, C, lor, Q, k, nprobe, "l2", id_base=100)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
    ix.close()


def test_candidate_rerank_on_a_multi_device_handle(vdb, oracle):
    """vdb_rerank on shards: global candidate ids are mapped to each shard's rows through its segment triples (appends
    included), partial lists merged -- equal to the single-device index."""
    import torch

    X, Q = _gauss(30_000, 40, 33, 17)
    multi = vdb.FlatIndex(40, "l2", [0, 0, 0])
    single = vdb.FlatIndex(40, "l2", 0)
    for a, b in ((0, 12_001), (12_001, 30_000)):            # two adds: every shard holds two id ranges
        multi.add(X[a:b], id_base=50)
        single.add(X[a:b], id_base=50)
    rng = np.random.default_rng(2)
    cand = np.stack([rng.choice(30_000, 64, replace=False) for _ in range(len(Q))]).astype(np.int64) + 50
    cand[:, 60:] = -1                                        # empty slots
    cand[0, :5] = [10, 49, 30_050 + 7, 10 ** 9, 50]          # ids outside the index (50 = row 0 is inside)
    for k in (1, 10, 64):
        Dm, Im = multi.rerank(Q, cand, k)
        Ds, Is = single.rerank(Q, cand, k)
        np.testing.assert_array_equal(Im, Is)
        np.testing.assert_array_equal(Dm, Ds)
    multi.set_option("multi_stage_all", 1)
    np.testing.assert_array_equal(multi.rerank(Q, cand, 10)[1], single.rerank(Q, cand, 10)[1])
    # device pointers
    from vdbhip import _ffi
    dev = torch.device("cuda", 0)
    qd, cd = torch.from_numpy(Q).to(dev), torch.from_numpy(cand).to(dev)
    Dd = torch.empty((len(Q), 10), dtype=torch.float32, device=dev)
    Id = torch.empty((len(Q), 10), dtype=torch.int64, device=dev)
    _ffi.check(_ffi.load().vdb_rerank_device(multi._handle(), qd.data_ptr(), len(Q), cd.data_ptr(), cand.shape[1], 10,
                                             Dd.data_ptr(), Id.data_ptr(), torch.cuda.current_stream().cuda_stream or None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(Id.cpu().numpy(), single.rerank(Q, cand, 10)[1])
    multi.close()
    single.close()
