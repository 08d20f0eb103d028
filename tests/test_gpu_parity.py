"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle and the golden vectors.

Bar: neighbour indices bit-exact; distances within 1e-4 relative (north_star) -- in practice they are
the correctly rounded float32 of the float64 value, so the checks below use 2e-6.
"""
from __future__ import annotations

import numpy as np
import pytest

from oracle import ref_semantics as rs

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # tolerance stated by BASELINE.json north_star for float distances


@pytest.fixture(scope="module")
def vdb():
    import vdbhip

    from vdbhip import _ffi

    assert _ffi.device_count() >= 1, "no MI355X visible"
    return vdbhip


def _kat(case):
    if case == "kat1":
        r = np.random.RandomState(0)
        return r.randn(1000, 16).astype(np.float32), r.randn(5, 16).astype(np.float32)
    if case == "kat2":
        return rs.random_dataset(128, 10000, 100, 42)
    if case == "kat3":
        return rs.random_dataset(64, 20000, 256, 7)
    raise KeyError(case)


def _composite(vdb, dim, metric):
    return vdb.CompositeAlgorithm(name=f"hip_{metric}", dimension=dim, metric=metric,
                                  indexer={"type": "HipBruteForceIndexer", "metric": metric},
                                  searcher={"type": "HipLinearSearcher", "metric": metric})


# ------------------------------------------------------------------------------------------------
# golden vectors produced by the reference implementation
# ------------------------------------------------------------------------------------------------
def test_reference_unit_known_answer(vdb, golden_dir):
    g = np.load(golden_dir / "kat0_ref_unit.npz")
    for metric in ("l2", "cosine", "ip"):
        algo = _composite(vdb, 2, metric)
        algo.build_index(g["X"])
        d, i = algo.batch_search(g["Q"], k=2)
        np.testing.assert_array_equal(i, g[f"I_{metric}"])
        np.testing.assert_allclose(d, g[f"D_{metric}"], rtol=RTOL, atol=1e-6)
        assert d.dtype == np.float32 and i.dtype == np.int64


@pytest.mark.parametrize("case,fname,k", [("kat1", "kat1_rs0_1000x16.npz", 3),
                                          ("kat2", "kat2_random_10000x128.npz", 10)])
@pytest.mark.parametrize("metric", ["l2", "cosine", "ip"])
def test_golden_linear_searcher_conventions(vdb, golden_dir, case, fname, k, metric):
    g = np.load(golden_dir / fname)
    X, Q = _kat(case)
    algo = _composite(vdb, X.shape[1], metric)
    algo.build_index(X)
    d, i = algo.batch_search(Q, k=k)
    np.testing.assert_array_equal(i, g[f"I_{metric}"])
    np.testing.assert_allclose(d, g[f"D_{metric}"], rtol=RTOL, atol=1e-6)
    # single-query API returns 1-D arrays (modular.py:332-334)
    d1, i1 = algo.search(Q[0], k=k)
    assert d1.shape == (k,) and i1.shape == (k,)
    np.testing.assert_array_equal(i1, g[f"I_{metric}"][0])


def test_golden_large_k_and_ground_truth(vdb, golden_dir, oracle):
    g = np.load(golden_dir / "kat3_smoke_20000x64_k100.npz")
    X, Q = _kat("kat3")
    algo = vdb.get_algorithm_instance("HipExactSearch", 64, name="exact_hip", metric="l2")
    algo.build_index(X)
    d, i = algo.batch_search(Q, k=100)
    d_o, i_o = oracle.knn(X, Q, 100, "l2")
    np.testing.assert_array_equal(i, i_o)                 # bit-exact against the canonical oracle
    np.testing.assert_array_equal(d, d_o)
    # against the reference's float32 output: identical except inside rounding-size tie bands
    from tests.helpers import tie_band_mismatch_report

    ka, kb = oracle.pair_keys(X, Q, i, "l2"), oracle.pair_keys(X, Q, g["I_l2"], "l2")
    assert tie_band_mismatch_report(i, g["I_l2"], ka, kb, band=1e-6) <= 2
    np.testing.assert_allclose(np.sqrt(d), g["D_l2"], rtol=RTOL)
    assert rs.recall_at_k(g["GT"], i, 100) == pytest.approx(1.0, abs=1e-4)


def test_edge_cases_golden(vdb, golden_dir):
    g = np.load(golden_dir / "edge_cases.npz")
    X, Q = g["X"], g["Q"]
    from tests.helpers import assert_same_neighbours_modulo_ties

    for metric in ("l2", "cosine", "ip"):
        algo = _composite(vdb, 2, metric)
        algo.build_index(X)
        for k in (3, 7):
            d, i = algo.batch_search(Q, k=k)
            Dg, Ig = g[f"D_{metric}_k{k}"], g[f"I_{metric}_k{k}"]
            assert d.dtype == np.float32 and i.dtype == np.int64 and d.shape == (2, k)
            if k == 7:  # k > N: +inf / -1 padding (modular.py:357-359, 382-384)
                assert np.all(i[:, 5:] == -1) and np.all(np.isinf(d[:, 5:])) and np.all(d[:, 5:] > 0)
                assert_same_neighbours_modulo_ties(i[:, :5], Ig[:, :5], d[:, :5], Dg[:, :5])
            else:
                np.testing.assert_allclose(d, Dg, rtol=RTOL, atol=1e-6)
            assert not np.isnan(d).any()
    # float64 + Fortran-ordered corpus and float64 queries are accepted (modular.py:114-118)
    r = np.random.RandomState(0)
    X1 = r.randn(1000, 16).astype(np.float32)
    Q1 = r.randn(5, 16).astype(np.float32)
    algo = _composite(vdb, 16, "l2")
    algo.build_index(np.asfortranarray(X1[:50].astype(np.float64)))
    d, i = algo.batch_search(Q1.astype(np.float64), k=4)
    np.testing.assert_array_equal(i, g["I_f64F"])
    np.testing.assert_allclose(d, g["D_f64F"], rtol=RTOL)


# ------------------------------------------------------------------------------------------------
# HIP path vs canonical oracle on seeded inputs: both kernels paths, both metrics
# ------------------------------------------------------------------------------------------------
CASES = [
    # n, d, nq, k, metric, kind
    (1, 8, 3, 1, "l2", "gauss"),
    (7, 3, 5, 4, "ip", "gauss"),
    (300, 50, 17, 10, "l2", "gauss"),
    (5000, 128, 64, 10, "ip", "gauss"),     # dense small-corpus MFMA path (N <= 8192, nq >= 64)
    (8000, 64, 200, 100, "l2", "gauss"),
    (1024, 128, 500, 128, "l2", "sift"),     # IVF-coarse shape: top-128 of 1024 centroids
    (700, 20, 64, 1, "ip", "glove"),
    (4097, 33, 9, 64, "l2", "gauss"),
    (3000, 200, 11, 5, "l2", "gauss"),      # small N: exhaustive exact kernel
    (2000, 768, 4, 10, "ip", "gauss"),
    (40000, 200, 100, 10, "l2", "gauss"),   # D > 128: K-loop MFMA scan
    (36000, 768, 80, 10, "ip", "gauss"),
    (50000, 384, 70, 5, "l2", "gauss"),
    (40000, 160, 64, 10, "ip", "gauss"),    # 192 padded dims: odd number of 64-dim K-steps
    (9000, 128, 500, 10, "l2", "gauss"),    # 8192 < N <= 15360, too few superbins for the scan: dense path
    (15000, 100, 70, 50, "ip", "gauss"),
    (20000, 64, 300, 10, "l2", "gauss"),    # 8192 < N < 32768: MFMA scan once nq*N >= 4e6 ...
    (20000, 64, 100, 10, "l2", "gauss"),    # ... exhaustive exact kernel below that
    (12000, 200, 400, 5, "ip", "gauss"),    # same rule on the K-loop path
    (20000, 64, 300, 50, "l2", "gauss"),    # large k on a mid-size corpus: 64-row bins as superbins (direct mode)
    (40000, 128, 200, 100, "ip", "gauss"),
    (100000, 100, 120, 100, "l2", "sift"),  # 128-row bins as superbins
    (100000, 384, 60, 200, "ip", "gauss"),  # same on the K-loop path: MS MARCO-subset shape, ground-truth k
    (50000, 200, 90, 64, "l2", "gauss"),
    (40000, 128, 200, 10, "l2", "gauss"),   # MFMA scan path
    (40000, 128, 200, 10, "ip", "gauss"),
    (50000, 50, 130, 10, "ip", "glove"),
    (33000, 64, 64, 1, "l2", "gauss"),
    (60000, 96, 3, 10, "ip", "gauss"),       # tiny batch, still the MFMA pipeline
    (70000, 16, 100, 20, "l2", "gauss"),
    (100000, 100, 96, 100, "l2", "gauss"),   # k too large for the bin select at this N: exact kernel
    (120000, 100, 96, 100, "l2", "gauss"),   # k = 100 through the MFMA scan
    (65536, 128, 256, 10, "l2", "sift"),    # integer-valued: fp16 scan exact, real ties
]


def _make(n, d, nq, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "sift":
        X = np.clip(np.rint(rng.gamma(0.6, 40.0, size=(n, d))), 0, 218).astype(np.float32)
        Q = np.clip(np.rint(rng.gamma(0.6, 40.0, size=(nq, d))), 0, 218).astype(np.float32)
    elif kind == "glove":
        X = (0.5 * rng.standard_normal((n, d))).astype(np.float32)
        Q = (0.5 * rng.standard_normal((nq, d))).astype(np.float32)
    else:
        X = rng.standard_normal((n, d)).astype(np.float32)
        Q = rng.standard_normal((nq, d)).astype(np.float32)
    return X, Q


@pytest.mark.parametrize("n,d,nq,k,metric,kind", CASES)
def test_flat_index_bit_exact_vs_oracle(vdb, oracle, n, d, nq, k, metric, kind):
    X, Q = _make(n, d, nq, kind, seed=n + d + k)
    idx = vdb.FlatIndex(d, metric, 0)
    idx.add(X)
    D, I = idx.search(Q, k)
    st = idx.stats()
    Do, Io = oracle.knn(X, Q, k, metric)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    # the bin select needs the superbins (two per >=512-row chunk) to outnumber k four to one
    span = 1024 if d > 128 else 512              # p16 panels (D > 128) use 1024-row spans
    npad = (n + span - 1) // span * span
    big_batch = npad > 8192 and nq * n >= 4_000_000
    std_geom = npad // 256 >= 16 and npad // 256 >= 4 * k
    # direct-bin mode (32-row-tile layout): 128- or 64-row bins as superbins when N/256 of them are too few for k
    direct = npad // 256 >= 16 and any(4 * k <= (npad // 256) * (256 // rows) <= 2048 for rows in (128, 64))
    if (n >= 32768 or big_batch) and k <= 1024 and (std_geom or direct):
        assert st["last_path_name"] == "mfma_scan", st
        assert st["last_fallback_queries"] == 0, st
    elif npad <= 15360 and nq >= 64 and d <= 128 and 2 * k <= n:
        assert st["last_path_name"] == "mfma_scan" and st["last_fallback_queries"] == 0, st   # dense path
    else:
        assert st["last_path_name"] == "exact_scan", st
    # the forced other path must give the same bits
    if n >= 8192 and k * 4 <= n // 512:
        idx.set_option("force_path", 2 if st["last_path_name"] == "exact_scan" else 1)
        D2, I2 = idx.search(Q, k)
        np.testing.assert_array_equal(I2, Io)
        np.testing.assert_array_equal(D2, Do)
    idx.close()


def test_duplicates_and_exact_ties_use_smaller_id(vdb, oracle):
    rng = np.random.default_rng(5)
    base = rng.integers(0, 6, size=(4000, 32)).astype(np.float32)      # many equal distances
    X = np.concatenate([base, base[:1500], base[:700]])                 # exact duplicates
    rng.shuffle(X, axis=0)
    X = np.tile(X, (7, 1))[:40000]
    Q = rng.integers(0, 6, size=(128, 32)).astype(np.float32)
    for metric in ("l2", "ip"):
        idx = vdb.FlatIndex(32, metric, 0)
        idx.add(X)
        D, I = idx.search(Q, 16)
        Do, Io = oracle.knn(X, Q, 16, metric)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
        assert idx.stats()["last_path_name"] == "mfma_scan"
        idx.close()


def test_work_list_overflow_takes_the_exhaustive_pass(vdb, oracle):
    X, Q = _make(40000, 64, 96, "gauss", 3)
    idx = vdb.FlatIndex(64, "l2", 0)
    idx.add(X)
    idx.set_option("list_cap", 1)          # every query overflows its candidate list
    D, I = idx.search(Q, 10)
    st = idx.stats()
    assert st["last_path_name"] == "mfma_scan" and st["last_fallback_queries"] == 96, st
    Do, Io = oracle.knn(X, Q, 10, "l2")
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    idx.close()


def test_zero_vectors_and_degenerate_inputs(vdb, oracle):
    X = np.zeros((33000, 16), np.float32)       # every distance ties: all queries must fall back, ids 0..k-1
    X[100] = 1.0
    Q = np.zeros((64, 16), np.float32)
    Q[1] = 1.0
    idx = vdb.FlatIndex(16, "l2", 0)
    idx.add(X)
    D, I = idx.search(Q, 5)
    Do, Io = oracle.knn(X, Q, 5, "l2")
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    assert I[0].tolist() == [0, 1, 2, 3, 4] and I[1, 0] == 100
    idx.close()


def test_fp16_scan_error_bound_holds(vdb):
    """|score - exact| <= eps for every (query,row): the premise of the exactness guard."""
    for kind, metric in (("gauss", "l2"), ("gauss", "ip"), ("sift", "l2"), ("glove", "ip")):
        X, Q = _make(4096, 128 if kind != "glove" else 50, 48, kind, 11)
        idx = vdb.FlatIndex(X.shape[1], metric, 0)
        idx.add(X)
        scores, eps, cs = idx.debug_scan_scores(Q, 0, 4096)
        X64, Q64 = X.astype(np.float64), Q.astype(np.float64)
        dots = Q64 @ X64.T
        exact = ((X64 * X64).sum(1)[None, :] - 2.0 * dots) if metric == "l2" else -dots
        err = np.abs(scores.astype(np.float64) / cs - exact)
        bound = (eps.astype(np.float64) / cs)[:, None]
        assert np.all(err <= bound), (kind, metric, float(err.max()), float(bound.min()))
        if kind == "sift":
            assert idx.stats()["corpus_fp16_exact"] == 1
            assert float(err.max()) == 0.0          # integer data: the fp16 MFMA scan is exact
        idx.close()


def test_search_errors_are_runtime_errors(vdb):
    algo = vdb.get_algorithm_instance("HipExactSearch", 4, name="e", metric="l2")
    with pytest.raises(RuntimeError, match="Index has not been built yet."):
        algo.batch_search(np.zeros((2, 4), np.float32), 3)
    algo.build_index(np.eye(4, dtype=np.float32))
    with pytest.raises(RuntimeError):
        algo.batch_search(np.zeros((2, 5), np.float32), 3)          # wrong dimension: never ValueError
    with pytest.raises(RuntimeError):
        algo.batch_search(np.zeros((2, 4), np.float32), 0)
    c = _composite(vdb, 2, "dot")
    c.build_index(np.zeros((3, 2), np.float32))
    with pytest.raises(ValueError, match="Unsupported metric 'dot'"):   # modular.py:387
        c.batch_search(np.zeros((1, 2), np.float32), 1)


def test_exact_search_metric_mapping_and_padding(vdb, oracle):
    """ExactSearch: 'cosine' means raw inner product, no normalisation (exact_search.py:23)."""
    X, Q = _make(500, 24, 9, "gauss", 2)
    X *= np.linspace(0.1, 3.0, 500, dtype=np.float32)[:, None]
    a = vdb.HipExactSearch("e", 24, metric="cosine")
    a.build_index(X)
    d, i = a.batch_search(Q, k=6)
    do, io = oracle.knn(X, Q, 6, "ip")
    np.testing.assert_array_equal(i, io)
    np.testing.assert_array_equal(d, do)
    assert np.all(np.diff(d, axis=1) <= 0)      # raw inner products, descending
    d, i = a.batch_search(Q, k=600)             # k > N: faiss padding
    assert np.all(i[:, 500:] == -1) and np.all(d[:, 500:] == -np.finfo(np.float32).max)
    d1, i1 = a.search(Q[0], k=6)
    np.testing.assert_array_equal(i1, io[0])
    assert a.get_operations()["ndis"] == 9 * 500 + 9 * 500
    assert a.get_memory_usage() > 0


def test_sharded_partials_merge_is_shard_count_invariant(vdb, oracle):
    torch = pytest.importorskip("torch")
    X, Q = _make(90000, 64, 128, "gauss", 9)
    k = 10
    dev = torch.device("cuda:0")
    q_t = torch.from_numpy(Q).to(dev)
    for metric in ("l2", "ip"):
        Do, Io = oracle.knn(X, Q, k, metric)
        for bounds in ([0, 90000], [0, 33000, 90000], [0, 100, 40000, 40001, 90000]):
            parts = len(bounds) - 1
            keys = torch.empty((parts, 128, k), dtype=torch.float64, device=dev)
            ids = torch.empty((parts, 128, k), dtype=torch.int64, device=dev)
            shards = []
            for p in range(parts):
                lo, hi = bounds[p], bounds[p + 1]
                s = vdb.FlatIndex(64, metric, 0)
                s.add(X[lo:hi], id_base=lo)
                s.search_partial_device(q_t.data_ptr(), 128, k, keys[p].data_ptr(), ids[p].data_ptr())
                shards.append(s)
            D = torch.empty((128, k), dtype=torch.float32, device=dev)
            I = torch.empty((128, k), dtype=torch.int64, device=dev)
            vdb.merge_partials_device(metric, 0, keys.data_ptr(), ids.data_ptr(), parts, 128, k, D.data_ptr(),
                                      I.data_ptr())
            torch.cuda.synchronize()
            np.testing.assert_array_equal(I.cpu().numpy(), Io)
            np.testing.assert_array_equal(D.cpu().numpy(), Do)
            # packed form: what ONE all-gather of [keys | ids] per rank produces
            packed = torch.stack([torch.stack([keys[p].view(torch.int64), ids[p]]) for p in range(parts)]).contiguous()
            D.zero_()
            I.zero_()
            vdb.merge_packed_partials_device(metric, 0, packed.data_ptr(), parts, 128, k, D.data_ptr(), I.data_ptr())
            torch.cuda.synchronize()
            np.testing.assert_array_equal(I.cpu().numpy(), Io)
            np.testing.assert_array_equal(D.cpu().numpy(), Do)
            for s in shards:
                s.close()


def test_many_queries_multi_batch_and_large_k(vdb, oracle):
    """nq above the library's internal 16384-query batch; k = 1000 (8 list slots per lane in the refine kernel)."""
    X, Q = _make(40000, 32, 40000, "gauss", 17)
    idx = vdb.FlatIndex(32, "l2", 0)
    idx.add(X)
    D, I = idx.search(Q, 3)
    sample = np.random.default_rng(0).choice(len(Q), 300, replace=False)
    Do, Io = oracle.knn(X, Q[sample], 3, "l2")
    np.testing.assert_array_equal(I[sample], Io)
    np.testing.assert_array_equal(D[sample], Do)
    D, I = idx.search(Q[:20], 1000)
    Do, Io = oracle.knn(X, Q[:20], 1000, "l2")
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    with pytest.raises(RuntimeError):
        idx.search(Q[:2], 4096)          # k above the supported 2048
    idx.close()


def test_sharded_algorithm_single_rank(vdb, oracle):
    """HipShardedExactSearch without an initialised process group behaves like one shard (world = 1)."""
    pytest.importorskip("torch")
    X, Q = _make(50000, 48, 100, "gauss", 4)
    a = vdb.get_algorithm_instance("HipShardedExactSearch", 48, name="sh", metric="ip", device=0)
    a.build_index(X)
    d, i = a.batch_search(Q, k=10)
    do, io = oracle.knn(X, Q, 10, "ip")
    np.testing.assert_array_equal(i, io)
    np.testing.assert_array_equal(d, do)
    assert a.shard == (0, 50000)


def test_candidate_rerank_matches_reference_loop(vdb):
    """FaissSearcher._batch_search_lsh_rerank (modular.py:483-532) restated in NumPy vs the batched HIP rerank."""
    rng = np.random.default_rng(12)
    X = rng.standard_normal((5000, 40)).astype(np.float32)
    Q = rng.standard_normal((37, 40)).astype(np.float32)
    C, k = 90, 7
    cand = np.stack([rng.choice(5000, C, replace=False) for _ in range(37)]).astype(np.int64)
    cand[rng.random(cand.shape) < 0.3] = -1              # FAISS pads missing candidates with -1
    cand[5] = -1                                          # a query without any candidate
    cand[6, 3:] = -1                                      # fewer candidates than k
    for metric in ("l2", "ip"):
        idx = vdb.FlatIndex(40, metric, 0)
        idx.add(X)
        d, i = vdb.rerank_candidates(idx, Q, cand, k, metric)
        assert d.dtype == np.float32 and i.dtype == np.int64
        for r in range(37):
            valid = cand[r][cand[r] >= 0]
            if metric == "l2":
                sc = np.sum((X[valid].astype(np.float64) - Q[r].astype(np.float64)) ** 2, axis=1)
            else:
                sc = -(X[valid].astype(np.float64) @ Q[r].astype(np.float64))
            order = np.lexsort((valid, sc))[:k]
            n = len(order)
            np.testing.assert_array_equal(i[r, :n], valid[order])
            want = np.sqrt(sc[order]) if metric == "l2" else sc[order]
            np.testing.assert_allclose(d[r, :n], want, rtol=2e-6, atol=1e-6)
            assert np.all(i[r, n:] == -1) and np.all(np.isinf(d[r, n:]))
        idx.close()


@pytest.mark.parametrize("d,metric,kind", [(128, "l2", "sift"), (100, "l2", "gauss"), (50, "ip", "glove"), (64, "ip", "gauss")])
def test_p16_panel_layout_for_small_dims_is_exact(vdb, oracle, d, metric, kind):
    """Option panel_layout = 2: 16-row-tile panels + scan16_kernel (v_mfma_f32_16x16x32_f16) for D <= 128 -- the
    measured-and-not-adopted alternative of DESIGN 4.3.  Same bins, ids and guard: bit-exact results, and the
    per-query error bound holds on its raw scores."""
    X, Q = _make(40000, d, 150, kind, 21)
    idx = vdb.FlatIndex(d, metric, 0)
    idx.set_option("panel_layout", 2)
    idx.add(X, id_base=5)
    for k in (1, 10, 37):
        D, I = idx.search(Q, k)
        Do, Io = oracle.knn(X, Q, k, metric, id_base=5)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
        assert idx.stats()["last_path_name"] == "mfma_scan"
    scores, eps, cs = idx.debug_scan_scores(Q[:32], 1024, 3000)
    X64, Q64 = X[1024:4024].astype(np.float64), Q[:32].astype(np.float64)
    dots = Q64 @ X64.T
    exact = ((X64 * X64).sum(1)[None, :] - 2.0 * dots) if metric == "l2" else -dots
    err = np.abs(scores.astype(np.float64) / cs - exact)
    assert np.all(err <= (eps.astype(np.float64) / cs)[:, None])
    idx.close()


@pytest.mark.parametrize("d,metric", [(200, "l2"), (384, "ip"), (768, "l2")])
def test_fp16_scan_error_bound_holds_on_the_kloop_layout(vdb, d, metric):
    """Same premise as above for D > 128 (p16 panels, K-loop scan): |raw MFMA score - exact| <= eps for every pair."""
    X, Q = _make(3000, d, 40, "gauss", 31 + d)
    X[::7] *= 3.0                       # uneven norms
    idx = vdb.FlatIndex(d, metric, 0)
    idx.add(X)
    scores, eps, cs = idx.debug_scan_scores(Q, 100, 2500)
    X64, Q64 = X[100:2600].astype(np.float64), Q.astype(np.float64)
    dots = Q64 @ X64.T
    exact = ((X64 * X64).sum(1)[None, :] - 2.0 * dots) if metric == "l2" else -dots
    err = np.abs(scores.astype(np.float64) / cs - exact)
    bound = (eps.astype(np.float64) / cs)[:, None]
    assert np.all(err <= bound), (d, metric, float(err.max()), float(bound.min()))
    assert float(err.max()) > 0.0       # (the hook really returns fp16-scan values, not exact ones)
    idx.close()


def test_search_statistics_are_summed_over_the_counter_shards(vdb):
    """vdb_stats after an MFMA-scan search: candidate quads / re-scanned bins / fallbacks are counted per query in
    sharded device counters (common.hpp stat_add) and summed on the host."""
    X, Q = _make(60000, 128, 700, "sift", 77)
    idx = vdb.FlatIndex(128, "l2", 0)
    idx.add(X)
    k = 10
    idx.search(Q, k)
    st = idx.stats()
    assert st["last_path_name"] == "mfma_scan" and st["last_nq"] == 700
    # every query needs at least ceil(k / 4) candidate quads (4 rows each) and stays within its work-list capacity
    assert 700 * 3 <= st["last_candidates"] <= 700 * (2 * k + 32 + 32)
    assert 0 <= st["last_rescan_bins"] <= 700 * 16 and st["last_fallback_queries"] == 0
    idx.close()


# ------------------------------------------------------------------------------------------------
# int8 scan copy (scan_i8.hpp): byte-valued integer corpora, chosen per batch on the device
# ------------------------------------------------------------------------------------------------
def _bytes_data(n, d, nq, window, seed, extreme=False):
    rng = np.random.default_rng(seed)
    lo, hi = (0, 255) if window == "u8" else (-128, 127)
    if extreme:      # rows / queries pinned to the ends of the window: largest |dot|, |bias| the accumulators can meet
        X = rng.choice(np.array([lo, hi], np.float32), size=(n, d))
        Q = rng.choice(np.array([lo, hi], np.float32), size=(nq, d))
        X[: n // 3] = rng.integers(lo, hi + 1, size=(n // 3, d)).astype(np.float32)
    else:
        X = np.clip(np.rint(rng.gamma(0.6, 40.0, size=(n, d))), 0, 255).astype(np.float32) + (0 if window == "u8" else -128)
        Q = np.clip(np.rint(rng.gamma(0.6, 40.0, size=(nq, d))), 0, 255).astype(np.float32) + (0 if window == "u8" else -128)
    return X, Q


@pytest.mark.parametrize("n,d,nq,k,metric,window,extreme", [
    (65536, 128, 256, 10, "l2", "u8", False),
    (70001, 128, 1000, 10, "ip", "u8", False),
    (40000, 64, 130, 10, "l2", "u8", False),       # two 32-dim k-steps
    (50000, 50, 77, 7, "l2", "s8", False),         # padded dims, s8 window
    (50000, 100, 600, 20, "ip", "s8", False),
    (120000, 96, 200, 100, "l2", "u8", False),     # k = 100 through the bins
    (40000, 128, 300, 10, "l2", "u8", True),       # extremes of the window: overflow margins of t' and of the packing
    (40000, 128, 300, 10, "ip", "s8", True),
    (40000, 128, 300, 10, "l2", "s8", True),
])
@pytest.mark.parametrize("shape", [16, 32])       # 16: v_mfma_i32_16x16x64_i8 on layout "x16" (default); 32: 32x32x32
def test_int8_scan_bit_exact_and_chosen_on_device(vdb, oracle, n, d, nq, k, metric, window, extreme, shape):
    X, Q = _bytes_data(n, d, nq, window, seed=n + d, extreme=extreme)
    idx = vdb.FlatIndex(d, metric, 0)
    idx.set_option("i8_shape", shape)
    idx.add(X, id_base=3)
    D, I = idx.search(Q, k)
    st = idx.stats()
    assert st["last_path_name"] == "mfma_scan" and st["has_i8_copy"] == 1 and st["scan_dtype"] == 1, st
    assert st["last_fallback_queries"] == 0 or extreme, st
    Do, Io = oracle.knn(X, Q, k, metric, id_base=3)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    # the fp16 scan of the same index (option panel_dtype = 1) agrees bit for bit
    idx.set_option("panel_dtype", 1)
    D1, I1 = idx.search(Q, k)
    assert idx.stats()["scan_dtype"] == 0
    np.testing.assert_array_equal(I1, I)
    np.testing.assert_array_equal(D1, D)
    idx.set_option("panel_dtype", 0)
    # select on quads (4 rows per candidate) instead of the default octs: same result (layout "x16" has octs only: no-op there)
    idx.set_option("i8_group", 4)
    D4, I4 = idx.search(Q, k)
    assert idx.stats()["scan_dtype"] == 1
    np.testing.assert_array_equal(I4, I)
    np.testing.assert_array_equal(D4, D)
    idx.set_option("i8_group", 8)
    # a batch with one non-integer value, or one value outside the byte window, is served by the fp16 scan
    for bad in (0.5, 300.0):
        Q2 = Q.copy()
        Q2[nq // 2, d // 2] = bad
        D2, I2 = idx.search(Q2, k)
        assert idx.stats()["scan_dtype"] == 0
        Do2, Io2 = oracle.knn(X, Q2, k, metric, id_base=3)
        np.testing.assert_array_equal(I2, Io2)
        np.testing.assert_array_equal(D2, Do2)
    # queries in the OTHER byte window still use the int8 scan (the windows of corpus and queries are independent)
    other = Q - 128 if window == "u8" else Q + 128
    D3, I3 = idx.search(other, k)
    assert idx.stats()["scan_dtype"] == 1
    Do3, Io3 = oracle.knn(X, other, k, metric, id_base=3)
    np.testing.assert_array_equal(I3, Io3)
    np.testing.assert_array_equal(D3, Do3)
    idx.close()


@pytest.mark.parametrize("n,d", [(1_000_000, 64), (700_000, 128)])
def test_int8_x16_launch_shapes_equal_the_32x32x32_scan(vdb, oracle, n, d):
    """Every launch shape of the 16x16x64 scan (1024- and 512-query tiles, 4- and 8-tile stages, 1 / 2 / 4-wave serving shapes with
    their staging rings) against the 32x32x32 scan of the same corpus, bit for bit, and a query sample against the oracle."""
    X, Q = _bytes_data(n, d, 4096, "u8", seed=d)
    a = vdb.FlatIndex(d, "l2", 0)
    a.add(X, id_base=0)
    b = vdb.FlatIndex(d, "l2", 0)
    b.set_option("i8_shape", 32)
    b.add(X, id_base=0)
    Db, Ib = b.search(Q, 10)
    assert b.stats()["scan_dtype"] == 1
    for variant in (3, 2, 1, 0):
        a.set_option("i8_variant", variant)
        Da, Ia = a.search(Q, 10)
        assert a.stats()["scan_dtype"] == 1 and a.stats()["last_path_name"] == "mfma_scan"
        np.testing.assert_array_equal(Ia, Ib)
        np.testing.assert_array_equal(Da, Db)
    a.set_option("i8_variant", 3)
    Do, Io = oracle.knn(X, Q[:24], 10, "l2")
    np.testing.assert_array_equal(Ib[:24], Io)
    np.testing.assert_array_equal(Db[:24], Do)
    for nq in (1, 64, 65, 128, 129, 256, 257, 777):       # 1 / 2 / 4 waves per workgroup, then the batch shape with padding waves
        for ring, nt in ((0, 0), (2, 1), (8, 0), (4, 1)):
            a.set_option("i8_ring", ring)
            a.set_option("i8_nt", nt)
            Da, Ia = a.search(Q[:nq], 10)
            assert a.stats()["scan_dtype"] == 1
            np.testing.assert_array_equal(Ia, Ib[:nq])
            np.testing.assert_array_equal(Da, Db[:nq])
    a.close()
    b.close()


def test_x16_tuning_options_do_not_change_results(vdb, oracle):
    """`f16_stage_tiles` (4- / 8-tile LDS stages of the fp16 x16 scan), `scan_prio` (issue priority of one half of a workgroup), and on
    a byte-valued corpus `scan_pair` (both scans in one launch / two launches) for integer and non-integer batches of every launch shape."""
    rng = np.random.default_rng(7)
    for d in (64, 128):
        X = rng.standard_normal((150_000, d)).astype(np.float32)
        Q = rng.standard_normal((1100, d)).astype(np.float32)
        idx = vdb.FlatIndex(d, "l2", 0)
        idx.add(X)
        D0, I0 = idx.search(Q, 10)
        assert idx.stats()["scan_shape"] == 16 and idx.stats()["last_path_name"] == "mfma_scan"
        Do, Io = oracle.knn(X, Q[:48], 10, "l2")
        np.testing.assert_array_equal(I0[:48], Io)
        np.testing.assert_array_equal(D0[:48], Do)
        for opt, vals in (("f16_stage_tiles", (4, 8, 0)), ("scan_prio", (1, 2, 0)), ("scan_pair", (0, 1))):
            for v in vals:
                idx.set_option(opt, v)
                D, I = idx.search(Q, 10)
                np.testing.assert_array_equal(I, I0)
                np.testing.assert_array_equal(D, D0)
        idx.close()
    for d in (64, 128):
        X, Q = _bytes_data(200_000, d, 1100, "u8", seed=d + 1)
        idx = vdb.FlatIndex(d, "l2", 0)
        idx.add(X)
        for Qb, dtype in ((Q, 1), ((Q + 0.5).astype(np.float32), 0)):
            for nq in (1, 100, 200, 1100):
                idx.set_option("scan_pair", 1)
                D1, I1 = idx.search(Qb[:nq], 10)
                assert idx.stats()["scan_dtype"] == dtype
                idx.set_option("scan_pair", 0)
                D2, I2 = idx.search(Qb[:nq], 10)
                np.testing.assert_array_equal(I1, I2)
                np.testing.assert_array_equal(D1, D2)
            Do, Io = oracle.knn(X, Qb[:40], 10, "l2")
            np.testing.assert_array_equal(I1[:40], Io)
            np.testing.assert_array_equal(D1[:40], Do)
        idx.close()


def test_int8_copy_only_for_byte_valued_corpora(vdb):
    rng = np.random.default_rng(0)
    for X, want in ((rng.standard_normal((40000, 64)).astype(np.float32), 0),
                    (rng.integers(0, 300, size=(40000, 64)).astype(np.float32), 0),        # integers beyond one byte
                    (rng.integers(-128, 128, size=(40000, 64)).astype(np.float32), 1),
                    (rng.integers(0, 256, size=(40000, 200)).astype(np.float32), 0)):      # D > 128: K-loop path
        idx = vdb.FlatIndex(X.shape[1], "l2", 0)
        idx.add(X)
        assert idx.stats()["has_i8_copy"] == want
        idx.close()


@pytest.mark.parametrize("kind", ["sift", "gauss"])
def test_more_queries_than_one_pass_statistics_accumulate(vdb, oracle, kind):
    """A call with more queries than one pass holds (16 384) is served in several passes: results are the same as the
    oracle's on a sample from every pass, and vdb_stats reports the counters of the WHOLE call."""
    X, _ = _make(70000, 128, 1, kind, 31)
    _, Q = _make(1, 128, 40000, kind, 32)
    idx = vdb.FlatIndex(128, "l2", 0)
    idx.add(X)
    D, I = idx.search(Q, 10)
    st = idx.stats()
    assert st["last_path_name"] == "mfma_scan" and st["last_nq"] == 40000
    assert st["scan_dtype"] == (1 if kind == "sift" else 0)
    assert st["last_candidates"] >= 40000 * 5, st            # ~10 groups per query over ALL three passes, not the last one
    sample = np.r_[0:8, 16380:16390, 32760:32776, 39990:40000]
    Do, Io = oracle.knn(X, Q[sample], 10, "l2")
    np.testing.assert_array_equal(I[sample], Io)
    np.testing.assert_array_equal(D[sample], Do)
    idx.close()


@pytest.mark.parametrize("kind,d,metric", [("sift", 128, "l2"), ("gauss", 128, "l2"), ("glove", 50, "ip"), ("sift", 64, "ip")])
def test_small_batches_use_narrow_workgroups_and_stay_exact(vdb, oracle, kind, d, metric):
    """Serving-shaped batches: up to 512 queries the scan takes finer row chunks, up to 256 queries 1 / 2 / 4-wave
    workgroups (search_flat.inc search_batch).  Every batch size on both sides of each threshold returns the oracle's bits,
    the same as the batch-shaped grid (`small_batch` = 0), for the int8 and the fp16 scan and both panel depths."""
    X, Q = _make(150_000, d, 2100, kind, 77)
    idx = vdb.FlatIndex(d, metric, 0)
    idx.add(X)
    Do, Io = oracle.knn(X, Q, 10, metric)
    for nq in (1, 2, 63, 64, 65, 128, 129, 256, 257, 512, 513, 2049):
        for sb in (1, 0):
            idx.set_option("small_batch", sb)
            D, I = idx.search(Q[:nq], 10)
            st = idx.stats()
            assert st["last_path_name"] == "mfma_scan" and st["last_fallback_queries"] == 0, (nq, sb, st)
            assert st["scan_dtype"] == (1 if kind == "sift" else 0)
            np.testing.assert_array_equal(I, Io[:nq], err_msg=f"nq={nq} small_batch={sb}")
            np.testing.assert_array_equal(D, Do[:nq], err_msg=f"nq={nq} small_batch={sb}")
    # a different window of queries at the narrowest shape, k large enough to need many superbins
    idx.set_option("small_batch", 1)
    D, I = idx.search(Q[1000:1003], 50)
    Do50, Io50 = oracle.knn(X, Q[1000:1003], 50, metric)
    np.testing.assert_array_equal(I, Io50)
    np.testing.assert_array_equal(D, Do50)
    idx.close()


@pytest.mark.parametrize("d,metric", [(768, "ip"), (384, "l2"), (200, "cosine"), (1600, "ip")])
def test_small_batches_on_the_kloop_scan_stay_exact(vdb, oracle, d, metric):
    """D > 128, serving-shaped batches: waves whose query columns are all padding only stage panels; below 64 queries the
    one active wave works on the column blocks that hold queries; up to 16 queries (D <= 1536) its B fragments live in
    LDS (scan16.hpp NARROW 1 / 2).  Batch sizes either side of every threshold against the oracle, and against the
    batch-shaped kernel (`small_batch` = 0)."""
    rng = np.random.default_rng(d)
    n = 40_000
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((130, d)).astype(np.float32)
    algo_metric = "ip" if metric == "cosine" else metric
    if metric == "cosine":
        X, Q = rs.safe_normalize(X), rs.safe_normalize(Q)
    idx = vdb.FlatIndex(d, algo_metric, 0)
    idx.add(X)
    Do, Io = oracle.knn(X, Q, 10, algo_metric)
    for nq in (1, 5, 16, 17, 33, 63, 64, 65, 130):
        for sb in (1, 0):
            idx.set_option("small_batch", sb)
            D, I = idx.search(Q[:nq], 10)
            st = idx.stats()
            assert st["last_path_name"] == "mfma_scan" and st["last_fallback_queries"] == 0, (nq, sb, st)
            np.testing.assert_array_equal(I, Io[:nq], err_msg=f"nq={nq} small_batch={sb}")
            np.testing.assert_array_equal(D, Do[:nq], err_msg=f"nq={nq} small_batch={sb}")
    idx.close()


def test_graph_replay_of_a_serving_loop_is_exact(vdb, oracle):
    """Option `graph`: a device-resident search repeated with the same buffers, shape and stream is captured into a
    hipGraph on its second call and replayed from the third.  The queries are rewritten IN PLACE between calls: every
    replay returns the oracle's bits for the queries of that call; a change of shape or options drops the graph."""
    import torch

    rng = np.random.default_rng(3)
    X = rng.standard_normal((90_000, 96)).astype(np.float32)
    Q = rng.standard_normal((6 * 24, 96)).astype(np.float32)
    Do, Io = oracle.knn(X, Q, 10, "l2")
    dev = torch.device("cuda", 0)
    idx = vdb.FlatIndex(96, "l2", 0)
    idx.add(X)
    idx.set_option("graph", 1)
    side = torch.cuda.Stream()
    q_t = torch.empty((24, 96), dtype=torch.float32, device=dev)
    D_t = torch.empty((24, 10), dtype=torch.float32, device=dev)
    I_t = torch.empty((24, 10), dtype=torch.int64, device=dev)
    for call in range(6):
        with torch.cuda.stream(side):
            q_t.copy_(torch.from_numpy(Q[24 * call:24 * call + 24]), non_blocking=False)
        idx.search_device(q_t.data_ptr(), 24, 10, D_t.data_ptr(), I_t.data_ptr(), side.cuda_stream)
        side.synchronize()
        np.testing.assert_array_equal(I_t.cpu().numpy(), Io[24 * call:24 * call + 24], err_msg=f"call {call}")
        np.testing.assert_array_equal(D_t.cpu().numpy(), Do[24 * call:24 * call + 24], err_msg=f"call {call}")
    st = idx.stats()
    assert st["graph_replays"] == 5, st                    # call 0 eager, call 1 captured + launched, 2..5 replayed
    idx.search_device(q_t.data_ptr(), 7, 10, D_t.data_ptr(), I_t.data_ptr(), side.cuda_stream)   # other shape: eager again
    side.synchronize()
    np.testing.assert_array_equal(I_t.cpu().numpy()[:7], Io[120:127])
    assert idx.stats()["graph_replays"] == 5
    # a larger batch grows the workspace: the buffers the graph was captured with are gone, so the old graph must not be
    # replayed when its shape comes back (it is re-captured instead)
    qb_t = torch.from_numpy(Q).to(dev)
    Db_t = torch.empty((len(Q), 10), dtype=torch.float32, device=dev)
    Ib_t = torch.empty((len(Q), 10), dtype=torch.int64, device=dev)
    idx.search_device(qb_t.data_ptr(), len(Q), 10, Db_t.data_ptr(), Ib_t.data_ptr(), side.cuda_stream)
    side.synchronize()
    np.testing.assert_array_equal(Ib_t.cpu().numpy(), Io)
    for call in range(3):
        with torch.cuda.stream(side):
            q_t.copy_(torch.from_numpy(Q[24 * call:24 * call + 24]))
        idx.search_device(q_t.data_ptr(), 24, 10, D_t.data_ptr(), I_t.data_ptr(), side.cuda_stream)
        side.synchronize()
        np.testing.assert_array_equal(I_t.cpu().numpy(), Io[24 * call:24 * call + 24], err_msg=f"after growth, call {call}")
    assert idx.stats()["graph_replays"] == 7                # eager, captured + launched, replayed
    # a buffer of this process moves BETWEEN two replays of the same key (another index is built and closed): the
    # allocation-epoch check must drop the captured graph -- it holds raw addresses -- and capture again
    other = vdb.FlatIndex(96, "l2", 0)
    other.add(X[:5000])
    other.search(Q[:8], 10)
    other.close()
    for call in range(3, 6):
        with torch.cuda.stream(side):
            q_t.copy_(torch.from_numpy(Q[24 * call:24 * call + 24]))
        idx.search_device(q_t.data_ptr(), 24, 10, D_t.data_ptr(), I_t.data_ptr(), side.cuda_stream)
        side.synchronize()
        np.testing.assert_array_equal(I_t.cpu().numpy(), Io[24 * call:24 * call + 24], err_msg=f"after epoch bump, call {call}")
        np.testing.assert_array_equal(D_t.cpu().numpy(), Do[24 * call:24 * call + 24], err_msg=f"after epoch bump, call {call}")
    assert idx.stats()["graph_replays"] == 9                # the stale exec is dropped: eager warm-up, re-captured + launched, replayed
    # a graph is dropped while its last launch may still be running: shape A twice (captured + launched), then shape B at
    # once on the same stream, no synchronisation in between (the exec is destroyed only behind its launch's event)
    D7_t = torch.empty((7, 10), dtype=torch.float32, device=dev)
    I7_t = torch.empty((7, 10), dtype=torch.int64, device=dev)
    with torch.cuda.stream(side):
        q_t.copy_(torch.from_numpy(Q[:24]))
    side.synchronize()
    for _ in range(4):
        idx.search_device(q_t.data_ptr(), 24, 10, D_t.data_ptr(), I_t.data_ptr(), side.cuda_stream)
        idx.search_device(q_t.data_ptr(), 24, 10, D_t.data_ptr(), I_t.data_ptr(), side.cuda_stream)
        idx.search_device(q_t.data_ptr(), 7, 10, D7_t.data_ptr(), I7_t.data_ptr(), side.cuda_stream)
    side.synchronize()
    np.testing.assert_array_equal(I_t.cpu().numpy(), Io[:24])
    np.testing.assert_array_equal(I7_t.cpu().numpy(), Io[:7])
    np.testing.assert_array_equal(D7_t.cpu().numpy(), Do[:7])
    idx.set_option("graph", 0)
    idx.close()

    # IVF: 18 dispatches per search in one graph
    C = X[rng.choice(len(X), 64, replace=False)].copy()
    ivf = vdb.IVFFlatIndex(96, 64, "l2", 0)
    ivf.set_centroids(C)
    ivf.add(X)
    ivf.set_nprobe(6)
    ivf.set_option("graph", 1)
    Dv, Iv = oracle.ivf_search(X, C, ivf.assignment(), Q, 10, 6, "l2")
    for call in range(5):
        with torch.cuda.stream(side):
            q_t.copy_(torch.from_numpy(Q[24 * call:24 * call + 24]))
        ivf.search_device(q_t.data_ptr(), 24, 10, D_t.data_ptr(), I_t.data_ptr(), side.cuda_stream)
        side.synchronize()
        np.testing.assert_array_equal(I_t.cpu().numpy(), Iv[24 * call:24 * call + 24], err_msg=f"ivf call {call}")
        np.testing.assert_array_equal(D_t.cpu().numpy(), Dv[24 * call:24 * call + 24], err_msg=f"ivf call {call}")
    assert ivf.stats()["graph_replays"] == 4
    ivf.close()


@pytest.mark.parametrize("d,metric,n", [(384, "ip", 150_000), (200, "l2", 70_000), (136, "l2", 1_600_000)])
def test_streamed_panels_halve_the_footprint_and_stay_exact(vdb, oracle, d, metric, n):
    """Option `stream_panels` (D > 128): the fp16 scan copy is not kept; every search converts the float32 rows slab by
    slab into one scratch slab (1.8x the corpus resident for a config-5 shard).  Results are the
    resident index's, bit for bit, for batch- and serving-shaped searches, and the footprint drops by the panel bytes."""
    rng = np.random.default_rng(d + n)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((300 if n < 1_000_000 else 64, d)).astype(np.float32)
    Do, Io = oracle.knn(X, Q, 10, metric)
    res = vdb.FlatIndex(d, metric, 0)
    res.add(X)
    Dr, Ir = res.search(Q, 10)
    np.testing.assert_array_equal(Ir, Io)
    resident = res.stats()["bytes_resident"]
    res.close()
    idx = vdb.FlatIndex(d, metric, 0)
    idx.set_option("stream_panels", 1)
    if n >= 1_000_000:
        idx.set_option("stream_slab_rows", 300_000)      # (several slabs at a test-sized corpus)
    idx.add(X)
    for nq in (len(Q), 64, 5, 1):
        D, I = idx.search(Q[:nq], 10)
        st = idx.stats()
        assert st["last_path_name"] == "mfma_scan" and st["last_fallback_queries"] == 0, st
        np.testing.assert_array_equal(I, Io[:nq], err_msg=f"nq={nq}")
        np.testing.assert_array_equal(D, Do[:nq], err_msg=f"nq={nq}")
    D, I = idx.search(Q, 10)
    np.testing.assert_array_equal(D, Dr)
    streamed = idx.stats()["bytes_resident"]
    dpad = -(-d // 64) * 64
    panels = n * dpad * 2
    if n >= 1_000_000:         # (a corpus of several slabs)
        assert streamed < resident - 0.4 * panels, (streamed, resident, panels)
    print(f"resident {resident / 2**20:.0f} MiB -> streamed {streamed / 2**20:.0f} MiB (corpus {X.nbytes / 2**20:.0f} MiB)")
    idx.close()
    if n < 100_000:            # the same through the plugin's config entry (`engine_options`), and a bad name fails loudly
        algo = vdb.HipExactSearch("exact_streamed", d, metric=metric, engine_options={"stream_panels": 1})
        algo.build_index(X)
        Dp, Ip = algo.batch_search(Q, 10)
        np.testing.assert_array_equal(Ip, Io)
        np.testing.assert_array_equal(Dp, Do)
        with pytest.raises(Exception):
            vdb.HipExactSearch("bad", d, metric=metric, engine_options={"no_such_option": 1}).build_index(X)


@pytest.mark.parametrize("d,metric,kind", [(128, "l2", "bytes"), (64, "l2", "gauss"), (96, "ip", "gauss"), (200, "l2", "gauss"),
                                           (384, "ip", "streamed")])
def test_add_appends_like_faiss(vdb, oracle, d, metric, kind):
    """`faiss.Index.add` appends; `vdb_add` used to replace the corpus on a second call.  Adds of
    uneven parts (host and device memory) give the index of the concatenated rows: ids continue, results bit-equal to the
    oracle on the whole corpus; the id base belongs to the index; `reset` empties it."""
    import torch

    n, nq, k = 70000, 130, 10
    rng = np.random.default_rng(d)
    if kind == "bytes":
        X = rng.integers(0, 256, (n, d)).astype(np.float32)
        Q = rng.integers(0, 256, (nq, d)).astype(np.float32)
    else:
        X = rng.standard_normal((n, d)).astype(np.float32)
        Q = rng.standard_normal((nq, d)).astype(np.float32)
    Do, Io = oracle.knn(X, Q, k, metric, id_base=300)
    idx = vdb.FlatIndex(d, metric, 0)
    if kind == "streamed":
        idx.set_option("stream_panels", 1)
    cuts = [0, 1, 20000, 20007, 52000, n]
    for j, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if j == 3:                                      # one part from device memory
            part = torch.from_numpy(X[a:b]).cuda()
            idx.add_device(part.data_ptr(), b - a, id_base=300)
            torch.cuda.synchronize()
        else:
            idx.add(X[a:b], id_base=300)
        assert idx.ntotal == b and idx.stats()["ntotal"] == b
        if j == 2:                                      # searches between appends see the rows so far
            D, I = idx.search(Q, k)
            Dp, Ip = oracle.knn(X[:b], Q, k, metric, id_base=300)
            np.testing.assert_array_equal(I, Ip)
            np.testing.assert_array_equal(D, Dp)
    idx.add(X[:0], id_base=300)
    D, I = idx.search(Q, k)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    st = idx.stats()
    assert st["last_path_name"] == "mfma_scan" and st["last_fallback_queries"] == 0, st
    if kind == "bytes":
        assert st["has_i8_copy"] == 1 and st["scan_dtype"] == 1
    with pytest.raises(ValueError, match="id base"):
        idx.add(X[:10], id_base=0)
    D2, I2 = idx.search(Q, k)                           # (the refused add left the index as it was)
    np.testing.assert_array_equal(I2, Io)
    idx.reset()
    assert idx.ntotal == 0
    with pytest.raises(RuntimeError, match="not been built"):
        idx.search(Q, k)
    idx.add(X[:3000], id_base=9)
    D3, I3 = idx.search(Q, k)
    Ds, Is = oracle.knn(X[:3000], Q, k, metric, id_base=9)
    np.testing.assert_array_equal(I3, Is)
    np.testing.assert_array_equal(D3, Ds)
    idx.close()


@pytest.mark.parametrize("kind", ["bytes", "gauss", "mixed"])
def test_serving_batches_take_their_statistics_inside_the_prep_kernel(vdb, oracle, kind):
    """Batches of <= 4096 query values skip the separate statistics dispatch: every workgroup of the prep kernel scans the
    batch itself (scan_i8.hpp, `fused_stats`).  Both settings give the oracle's bits on every scan the statistics choose
    between: int8 (byte-valued integer queries), fp16 (real values), exhaustive (a non-finite value)."""
    n, d, k = 60000, 128, 10
    rng = np.random.default_rng(77)
    X = rng.integers(0, 256, (n, d)).astype(np.float32) if kind != "gauss" else rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.integers(0, 256, (40, d)).astype(np.float32) if kind != "gauss" else rng.standard_normal((40, d)).astype(np.float32)
    if kind == "mixed":
        Q[3, 5] += 0.5                                   # one non-integer value: the batches that hold it take the fp16 scan
    Do, Io = oracle.knn(X, Q, k, "l2")
    idx = vdb.FlatIndex(d, "l2", 0)
    idx.add(X)
    for fused in (1, 0):
        idx.set_option("fused_stats", fused)
        for nq in (1, 3, 4, 32, 33, 40):                 # 32 x 128 = 4096 values is the last fused size
            D, I = idx.search(Q[:nq], k)
            st = idx.stats()
            assert st["last_path_name"] == "mfma_scan" and st["last_fallback_queries"] == 0, (fused, nq, st)
            want_i8 = kind == "bytes" or (kind == "mixed" and nq <= 3)
            assert st["scan_dtype"] == (1 if want_i8 else 0), (fused, nq, st)
            np.testing.assert_array_equal(I, Io[:nq], err_msg=f"fused={fused} nq={nq}")
            np.testing.assert_array_equal(D, Do[:nq], err_msg=f"fused={fused} nq={nq}")
        Qbad = Q[:5].copy()
        Qbad[2, 7] = np.inf                              # unusable scales: every query of the batch goes the exhaustive way
        D, I = idx.search(Qbad, k)
        assert idx.stats()["last_fallback_queries"] == 5
        np.testing.assert_array_equal(I[[0, 1, 3, 4]], Io[[0, 1, 3, 4]])
    idx.close()
