"""BASELINE.json full sizes (1M x 128, 10k queries): size-independent properties + sampled oracle parity.

The oracle cannot brute-force 10^10 pairs in test time, so at full size the HIP path is checked through
 * a sampled exact comparison (64 queries against all 1M rows, bit-exact),
 * self-retrieval: queries copied from corpus rows come back first at distance 0 (L2) / self-score (IP),
 * sortedness and id validity of every row of the (10k, 10) result,
 * shard-count invariance: 1 shard == merge of 3 unequal shards, bit for bit,
 * idempotence: the same batch twice gives the same bits; a permuted batch gives permuted rows.
"""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vdb():
    import vdbhip

    return vdbhip


@pytest.fixture(scope="module")
def sift(vdb):
    from vdbhip import datasets

    return datasets.sift_like(1_000_000, 10_000, 128, 1234)


def _properties(D, I, n, metric):
    assert D.shape == I.shape and I.dtype == np.int64 and D.dtype == np.float32
    assert I.min() >= 0 and I.max() < n
    steps = np.diff(D, axis=1)
    assert np.all(steps >= 0) if metric == "l2" else np.all(steps <= 0)
    srt = np.sort(I, axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1]), "duplicate neighbour ids in a row"
    ties = steps == 0                              # equal distances must be ordered by id
    assert np.all(np.diff(I, axis=1)[ties] > 0)


def test_sift1m_full_batch(vdb, sift, oracle):
    X, Q = sift
    idx = vdb.FlatIndex(128, "l2", 0)
    idx.add(X)
    D, I = idx.search(Q, 10)
    st = idx.stats()
    assert st["last_path_name"] == "mfma_scan" and st["corpus_fp16_exact"] == 1
    assert st["last_fallback_queries"] == 0
    _properties(D, I, len(X), "l2")
    sample = np.random.default_rng(0).choice(len(Q), 64, replace=False)
    Do, Io = oracle.knn(X, Q[sample], 10, "l2")
    np.testing.assert_array_equal(I[sample], Io)
    np.testing.assert_array_equal(D[sample], Do)
    # idempotence and batch-order independence
    D2, I2 = idx.search(Q, 10)
    np.testing.assert_array_equal(I2, I)
    np.testing.assert_array_equal(D2, D)
    perm = np.random.default_rng(1).permutation(len(Q))
    D3, I3 = idx.search(Q[perm], 10)
    np.testing.assert_array_equal(I3, I[perm])
    np.testing.assert_array_equal(D3, D[perm])
    # self retrieval (SIFT-like data has exact duplicates: the smallest id of the duplicate group wins)
    rows = np.random.default_rng(2).choice(len(X), 2000, replace=False)
    Ds, Is = idx.search(X[rows], 10)
    assert np.all(Ds[:, 0] == 0)
    assert np.all(Is[:, 0] <= rows)
    assert np.all((X[Is[:, 0]] == X[rows]).all(axis=1))
    idx.close()


def test_sift1m_shard_invariance(vdb, sift):
    torch = pytest.importorskip("torch")
    X, Q = sift
    Q = Q[:2048]
    k = 10
    dev = torch.device("cuda:0")
    q_t = torch.from_numpy(Q).to(dev)
    full = vdb.FlatIndex(128, "l2", 0)
    full.add(X)
    D1, I1 = full.search(Q, k)
    full.close()
    bounds = [0, 300_001, 777_777, 1_000_000]
    parts = len(bounds) - 1
    keys = torch.empty((parts, len(Q), k), dtype=torch.float64, device=dev)
    ids = torch.empty((parts, len(Q), k), dtype=torch.int64, device=dev)
    shards = []
    for p in range(parts):
        s = vdb.FlatIndex(128, "l2", 0)
        s.add(X[bounds[p]:bounds[p + 1]], id_base=bounds[p])
        s.search_partial_device(q_t.data_ptr(), len(Q), k, keys[p].data_ptr(), ids[p].data_ptr())
        shards.append(s)
    D = torch.empty((len(Q), k), dtype=torch.float32, device=dev)
    I = torch.empty((len(Q), k), dtype=torch.int64, device=dev)
    vdb.merge_partials_device("l2", 0, keys.data_ptr(), ids.data_ptr(), parts, len(Q), k, D.data_ptr(), I.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(I.cpu().numpy(), I1)
    np.testing.assert_array_equal(D.cpu().numpy(), D1)
    for s in shards:
        s.close()


def test_gaussian1m_and_glove_shapes(vdb, oracle):
    """configs[1] Gaussian variant and configs[2] (GloVe-50 shape, inner product) with the full 10 000-query batch."""
    from vdbhip import datasets

    for (X, Q), metric in ((datasets.gaussian(1_000_000, 10_000, 128, 1234), "l2"),
                           (datasets.glove_like(1_200_000, 10_000, 50, 50), "ip")):
        idx = vdb.FlatIndex(X.shape[1], metric, 0)
        idx.add(X)
        D, I = idx.search(Q, 10)
        st = idx.stats()
        assert st["last_path_name"] == "mfma_scan" and st["last_fallback_queries"] == 0, st
        _properties(D, I, len(X), metric)
        sample = np.random.default_rng(3).choice(len(Q), 48, replace=False)
        Do, Io = oracle.knn(X, Q[sample], 10, metric)
        np.testing.assert_array_equal(I[sample], Io)
        np.testing.assert_array_equal(D[sample], Do)
        # the other launch forms of the x16 fp16 scan on the full batch: 512-query tiles (D <= 64 takes 1024-query tiles by default),
        # 4- / 8-tile stages, and the 32x32x16 kernel on its own layout -- all bit-identical
        assert st["scan_shape"] == 16
        for opt, v in (("f16_wide", 1), ("f16_stage_tiles", 4), ("f16_stage_tiles", 8)):
            idx.set_option(opt, v)
            D2, I2 = idx.search(Q, 10)
            np.testing.assert_array_equal(I2, I)
            np.testing.assert_array_equal(D2, D)
        idx.close()
        old = vdb.FlatIndex(X.shape[1], metric, 0)
        old.set_option("flat_shape", 32)
        old.add(X)
        D3, I3 = old.search(Q, 10)
        assert old.stats()["scan_shape"] == 32
        np.testing.assert_array_equal(I3, I)
        np.testing.assert_array_equal(D3, D)
        old.close()


def test_glove_shape_cosine_through_the_plugin(vdb, oracle):
    """configs[2] cosine variant at full size through BruteForceIndexer + LinearSearcher semantics
    (modular.py:315-325, 363-385: both sides normalised, negated scores ascending)."""
    from oracle import ref_semantics as rs
    from vdbhip import datasets

    X, Q = datasets.glove_like(1_200_000, 10_000, 50, 50)
    algo = vdb.CompositeAlgorithm("exact_cos", 50, indexer={"type": "HipBruteForceIndexer", "metric": "cosine"},
                                  searcher={"type": "HipLinearSearcher", "metric": "cosine"}, metric="cosine")
    algo.build_index(X)
    d, i = algo.batch_search(Q, k=10)
    assert d.dtype == np.float32 and i.dtype == np.int64 and d.shape == (10_000, 10)
    assert np.all(np.diff(d, axis=1) >= 0) and np.all(d <= 0) and np.all(d >= -1.0001)
    assert i.min() >= 0 and i.max() < len(X)
    sample = np.random.default_rng(5).choice(len(Q), 48, replace=False)
    Xn, Qn = rs.safe_normalize(X), rs.safe_normalize(Q[sample])
    Do, Io = oracle.knn(Xn, Qn, 10, "ip")
    do, io = rs.flat_to_linear(Do, Io, "cosine")
    np.testing.assert_array_equal(i[sample], io)
    np.testing.assert_allclose(d[sample], do, rtol=1e-4, atol=1e-6)     # north_star: distances within 1e-4 relative
