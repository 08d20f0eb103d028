"""Option `int8_only`: a byte-valued corpus keeps ONLY its int8 copies (the reference holds one copy of the corpus,
exact_search.py:34-39).  Same results as the default index and the oracle for integer AND non-integer query batches, through
every kernel family that used to read the float32 rows (work-list refine, exhaustive scans, rerank, reserve, fallback)."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vdb():
    import vdbhip

    return vdbhip


def _bytes(n, d, seed, lo=0, hi=218):
    rng = np.random.default_rng(seed)
    return np.clip(np.round(rng.gamma(0.6, 40.0, size=(n, d))) + lo, lo, hi).astype(np.float32)


def _index(vdb, X, metric, devices=0, **opts):
    ix = vdb.FlatIndex(X.shape[1], metric, devices)
    ix.set_option("int8_only", 1)
    for k, v in opts.items():
        ix.set_option(k, v)
    ix.add(X, id_base=9)
    return ix


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("n,d,window", [(70_001, 128, "u8"), (50_000, 64, "u8"), (60_000, 100, "s8"), (40_000, 50, "u8")])
def test_int8_only_equals_default_index_and_oracle(vdb, oracle, metric, n, d, window):
    X = _bytes(n, d, n + d) if window == "u8" else _bytes(n, d, n + d, lo=-128, hi=90)
    Qi = _bytes(300, d, 5) if window == "u8" else _bytes(300, d, 5, lo=-128, hi=90)        # integer queries: int8 scan
    Qf = (Qi + np.random.default_rng(6).standard_normal(Qi.shape).astype(np.float32) * 3).astype(np.float32)   # fp16 slabs
    ix = _index(vdb, X, metric, int8_block_rows=16_384, i8_shape=32 if d == 64 else 0)   # (several ingestion blocks; one case on the 32x32x32 panels)
    st = ix.stats()
    assert st["has_i8_copy"] == 2 and st["ntotal"] == n
    corpus = X.nbytes
    assert (st["bytes_resident"] - st["bytes_workspace"]) <= 0.62 * max(corpus, n * 128 * 4), st
    ref = vdb.FlatIndex(d, metric, 0)
    ref.add(X, id_base=9)
    for Q, want_dtype in ((Qi, 1), (Qf, 0)):
        for k in (1, 10, 100):
            D, I = ix.search(Q, k)
            Dr, Ir = ref.search(Q, k)
            np.testing.assert_array_equal(I, Ir)
            np.testing.assert_array_equal(D, Dr)
        Do, Io = oracle.knn(X, Q, 10, metric, id_base=9)
        D, I = ix.search(Q, 10)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
        s = ix.stats()           # (k = 100 on these few rows takes the finer fp16 bins whatever the queries: checked at k = 10)
        assert s["last_path_name"] == "mfma_scan" and s["scan_dtype"] == want_dtype, s
    # single queries / small batches (serving shapes), and the exhaustive kernels on the int8 rows
    for Q in (Qi[:1], Qf[:3], Qf[:70]):
        np.testing.assert_array_equal(ix.search(Q, 10)[1], oracle.knn(X, Q, 10, metric, id_base=9)[1])
    for fp in (1, 3):
        ix.set_option("force_path", fp)
        D, I = ix.search(Qf[:40], 10)
        Do, Io = oracle.knn(X, Qf[:40], 10, metric, id_base=9)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
        assert ix.stats()["last_path_name"] == "exact_scan"
    ix.set_option("force_path", 0)
    # work lists that overflow -> the device-split exhaustive fallback, on int8 rows
    ix.set_option("list_cap", 1)
    D, I = ix.search(Qf[:50], 10)
    np.testing.assert_array_equal(I, oracle.knn(X, Qf[:50], 10, metric, id_base=9)[1])
    assert ix.stats()["last_fallback_queries"] > 0
    ix.set_option("list_cap", 0)
    # candidate re-scoring and reserve
    cand = np.random.default_rng(1).integers(0, n, size=(20, 37)).astype(np.int64)
    cand = np.stack([np.unique(r)[:30] for r in cand])
    Dc, Ic = ix.rerank(Qf[:20], cand + 9, 5)
    Drc, Irc = ref.rerank(Qf[:20], cand + 9, 5)
    np.testing.assert_array_equal(Ic, Irc)
    np.testing.assert_array_equal(Dc, Drc)
    ix.reserve(2000, 10)
    np.testing.assert_array_equal(ix.search(Qi, 10)[1], oracle.knn(X, Qi, 10, metric, id_base=9)[1])
    with pytest.raises(RuntimeError, match="ONE add"):
        ix.add(X[:100], id_base=9)
    ix.reset()
    ix.add(X[:40_000], id_base=0)
    assert ix.stats()["has_i8_copy"] == 2
    np.testing.assert_array_equal(ix.search(Qi[:9], 5)[1], oracle.knn(X[:40_000], Qi[:9], 5, metric)[1])
    ix.close()
    ref.close()


def test_window_restart_and_fallback_to_the_default_layout(vdb, oracle):
    """The byte window comes from the first ingestion block; a later block that leaves it restarts the build with the other
    window when that fits the whole corpus, and a corpus that is not byte-valued gets the default layout."""
    X = _bytes(60_000, 32, 3, lo=0, hi=100)
    X[50_000:] -= 90.0                       # rows of the last blocks go negative: s8 fits everything, u8 (block 0) does not
    Q = _bytes(64, 32, 4, lo=0, hi=100)
    ix = _index(vdb, X, "l2", int8_block_rows=8_192)
    assert ix.stats()["has_i8_copy"] == 2
    np.testing.assert_array_equal(ix.search(Q, 10)[1], oracle.knn(X, Q, 10, "l2", id_base=9)[1])
    ix.close()
    Xg = np.random.default_rng(0).standard_normal((40_000, 32)).astype(np.float32)
    ig = _index(vdb, Xg, "l2")
    assert ig.stats()["has_i8_copy"] == 0 and ig.stats()["ntotal"] == 40_000
    np.testing.assert_array_equal(ig.search(Xg[:8] + 0.01, 5)[1], oracle.knn(Xg, Xg[:8] + 0.01, 5, "l2", id_base=9)[1])
    ig.close()
    Xm = _bytes(40_000, 32, 8)
    Xm[39_999, 5] = 0.5                      # one non-integer value in the LAST block
    im = _index(vdb, Xm, "l2", int8_block_rows=8_192)
    assert im.stats()["has_i8_copy"] == 0
    np.testing.assert_array_equal(im.search(Q, 10)[1], oracle.knn(Xm, Q, 10, "l2", id_base=9)[1])
    im.close()


def test_int8_only_over_several_devices_and_through_the_plugin(vdb, oracle):
    X = _bytes(150_000, 128, 21)
    Q = _bytes(200, 128, 22)
    ix = _index(vdb, X, "l2", devices=[0, 0, 0])
    st = ix.stats()
    assert st["has_i8_copy"] == 2 and st["ndevices"] == 3
    np.testing.assert_array_equal(ix.search(Q, 10)[1], oracle.knn(X, Q, 10, "l2", id_base=9)[1])
    np.testing.assert_array_equal(ix.search(Q + 0.25, 10)[1], oracle.knn(X, Q + 0.25, 10, "l2", id_base=9)[1])
    ix.close()
    algo = vdb.get_algorithm_instance("HipExactSearch", 128, name="e", metric="l2", engine_options={"int8_only": 1})
    algo.build_index(X)
    assert algo.index.stats()["has_i8_copy"] == 2
    D, I = algo.batch_search(Q, 10)
    Do, Io = oracle.knn(X, Q, 10, "l2")
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert algo.get_memory_usage() * 2 ** 20 < 0.9 * X.nbytes + algo.index.stats()["bytes_workspace"]


def test_sift100m_shard_on_one_gpu():
    """100M x 128 byte-valued rows (12.8 GB of int8 rows + 12.8 GB of int8 panels; 51 GB as float32) generated on the device in
    blocks, built through vdb_add_device with `int8_only`, searched, and a query sample checked against a float64 torch scan."""
    import torch

    import bench
    import vdbhip

    dev = torch.device("cuda", 0)
    n, d, nq, k = 100_000_000, 128, 2_000, 10
    X_t = bench.device_byte_rows(n, d, dev, 31)
    ix = vdbhip.FlatIndex(d, "l2", 0)
    ix.set_option("int8_only", 1)
    ix.add_device(X_t.data_ptr(), n, id_base=0)
    torch.cuda.synchronize()
    st = ix.stats()
    assert st["has_i8_copy"] == 2 and st["ntotal"] == n
    index_bytes = st["bytes_resident"] - st["bytes_workspace"]
    assert index_bytes < 0.6 * n * d * 4, index_bytes            # 0.55 x the float32 corpus
    q_t = bench.device_byte_rows(nq, d, dev, 32)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s = ix.stats()
    assert s["last_path_name"] == "mfma_scan" and s["scan_dtype"] == 1 and s["last_fallback_queries"] == 0
    assert bench.device_check(X_t, q_t, I_t, k, "l2", 0, sample=16) == 1.0
    ids = I_t.cpu().numpy()
    assert ids.min() >= 0 and ids.max() < n and all(len(set(r)) == k for r in ids[:200].tolist())
    ix.close()
