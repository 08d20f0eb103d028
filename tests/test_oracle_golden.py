"""Pin the CPU oracle against golden vectors produced by the REFERENCE implementation.

The fixtures in tests/golden/ are outputs of the reference's own NumPy hot path
(BruteForceIndexer + LinearSearcher through CompositeAlgorithm, src/algorithms/modular.py:121-133,
312-390, 554-622) and of its dataset/metric helpers; see tests/golden/make_golden.py.
"""
from __future__ import annotations

import json

import numpy as np
import pytest

from oracle import ref_semantics as rs
from tests.helpers import assert_same_neighbours_modulo_ties


def _kat1():
    r = np.random.RandomState(0)
    return r.randn(1000, 16).astype(np.float32), r.randn(5, 16).astype(np.float32)


def _inputs(case):
    if case == "kat1":
        return _kat1()
    if case == "kat2":
        return rs.random_dataset(128, 10000, 100, 42)
    if case == "kat3":
        return rs.random_dataset(64, 20000, 256, 7)
    raise KeyError(case)


def _norm_for(metric, X, Q):
    """cosine = normalise both sides then inner product (modular.py:321-323, 364-366)."""
    if metric == "cosine":
        return rs.safe_normalize(X), rs.safe_normalize(Q), "ip"
    return X, Q, metric


def test_manifest_input_hashes(golden_dir):
    import hashlib

    man = json.loads((golden_dir / "manifest.json").read_text())
    for case, name in (("kat1", "kat1_rs0_1000x16"), ("kat2", "kat2_random_10000x128"),
                       ("kat3", "kat3_smoke_20000x64_k100")):
        X, Q = _inputs(case)
        assert hashlib.sha256(X.tobytes()).hexdigest() == man["cases"][name]["sha_X"]
        assert hashlib.sha256(Q.tobytes()).hexdigest() == man["cases"][name]["sha_Q"]
    # SURVEY 8c recorded the first 16 hex digits of the KAT-2 hashes
    assert man["cases"]["kat2_random_10000x128"]["sha_X"].startswith("20869e197d334376")
    assert man["cases"]["kat2_random_10000x128"]["sha_Q"].startswith("4e9303208848cbfc")


def test_reference_unit_known_answer(golden_dir, oracle):
    """tests/test_composite_algorithm.py:29-58 of the reference: 4x2 corpus, 2 queries, k=2."""
    g = np.load(golden_dir / "kat0_ref_unit.npz")
    X, Q = g["X"], g["Q"]
    expected = np.argsort(np.linalg.norm(X[None] - Q[:, None], axis=2), axis=1)[:, :2]
    np.testing.assert_array_equal(g["I_l2"], expected)
    for metric in ("l2", "cosine", "ip"):
        d_np, i_np = rs.linear_searcher_batch(X, Q, 2, metric)
        np.testing.assert_array_equal(i_np, g[f"I_{metric}"])
        np.testing.assert_allclose(d_np, g[f"D_{metric}"], rtol=1e-6, atol=1e-7)
        Xn, Qn, m = _norm_for(metric, X, Q)
        d_c, i_c = oracle.knn(Xn, Qn, 2, m)
        d_c, i_c = rs.flat_to_linear(d_c, i_c, m)
        np.testing.assert_array_equal(i_c, g[f"I_{metric}"])
        np.testing.assert_allclose(d_c, g[f"D_{metric}"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case,fname,k", [("kat1", "kat1_rs0_1000x16.npz", 3),
                                          ("kat2", "kat2_random_10000x128.npz", 10)])
@pytest.mark.parametrize("metric", ["l2", "cosine", "ip"])
def test_canonical_oracle_matches_reference_outputs(golden_dir, oracle, case, fname, k, metric):
    g = np.load(golden_dir / fname)
    X, Q = _inputs(case)
    Xn, Qn, m = _norm_for(metric, X, Q)
    d, i = oracle.knn(Xn, Qn, k, m, mode=oracle.MODE_CANON)
    d, i = rs.flat_to_linear(d, i, m)
    np.testing.assert_array_equal(i, g[f"I_{metric}"])
    np.testing.assert_allclose(d, g[f"D_{metric}"], rtol=2e-6, atol=1e-6)


def test_survey_known_answers(golden_dir):
    """Values recorded in SURVEY.md 8c during the survey session."""
    g1 = np.load(golden_dir / "kat1_rs0_1000x16.npz")
    assert g1["I_l2"][0].tolist() == [414, 652, 507]
    assert g1["I_cosine"][0].tolist() == [983, 414, 507]
    assert g1["I_ip"][0].tolist() == [752, 983, 311]
    np.testing.assert_allclose(g1["D_l2"][0], [3.5341165, 3.6194804, 3.7280023], rtol=1e-6)
    g2 = np.load(golden_dir / "kat2_random_10000x128.npz")
    assert g2["I_l2"][0].tolist() == [2689, 3260, 321, 3219, 4921, 4391, 7865, 5264, 8681, 6306]
    np.testing.assert_array_equal(g2["I_l2"], g2["GT"][:, :10].astype(np.int64))
    assert g2["I_cosine"][0].tolist() == [4921, 8572, 3707, 6667, 9886, 3416, 8681, 5733, 1302, 8149]
    assert g2["I_ip"][0].tolist() == [9886, 8572, 332, 3416, 1302, 3707, 4921, 6667, 5733, 7561]


@pytest.mark.parametrize("case,fname,k", [("kat1", "kat1_rs0_1000x16.npz", 3),
                                          ("kat2", "kat2_random_10000x128.npz", 10),
                                          ("kat3", "kat3_smoke_20000x64_k100.npz", 100)])
def test_numpy32_mode_is_bit_faithful_to_reference_l2(golden_dir, oracle, case, fname, k):
    """The float32 NumPy-order restatement reproduces the reference's L2 distances BIT FOR BIT."""
    g = np.load(golden_dir / fname)
    X, Q = _inputs(case)
    d, i = oracle.knn(X, Q, k, "l2", mode=oracle.MODE_NUMPY32)
    np.testing.assert_array_equal(i, g["I_l2"])
    assert np.array_equal(np.sqrt(d), g["D_l2"]), "sqrt(float32 pairwise sum) must equal the reference bits"


def test_large_k_golden_canonical(golden_dir, oracle):
    g = np.load(golden_dir / "kat3_smoke_20000x64_k100.npz")
    X, Q = _inputs("kat3")
    d, i = oracle.knn(X, Q, 100, "l2")
    # float32 (reference) and float64 (canonical) may swap neighbours only inside a rounding-size tie band
    if not np.array_equal(i, g["I_l2"]):
        ka = oracle.pair_keys(X, Q, i, "l2")
        kb = oracle.pair_keys(X, Q, g["I_l2"], "l2")
        from tests.helpers import tie_band_mismatch_report

        rows = tie_band_mismatch_report(i, g["I_l2"], ka, kb, band=1e-6)
        assert rows <= 2
    np.testing.assert_allclose(np.sqrt(d), g["D_l2"], rtol=2e-6)
    np.testing.assert_array_equal(g["I_l2"], g["GT"].astype(np.int64))


def test_gemm32_mode_matches_on_well_separated_fixture(golden_dir, oracle):
    g = np.load(golden_dir / "kat2_random_10000x128.npz")
    X, Q = _inputs("kat2")
    d, i = oracle.knn(X, Q, 10, "l2", mode=oracle.MODE_GEMM32)
    np.testing.assert_array_equal(i, g["I_l2"])
    np.testing.assert_allclose(np.sqrt(d), g["D_l2"], rtol=1e-5)
    d, i = oracle.knn(X, Q, 10, "ip", mode=oracle.MODE_GEMM32)
    np.testing.assert_array_equal(i, g["I_ip"])
    np.testing.assert_allclose(-d, g["D_ip"], rtol=1e-5)


def test_edge_cases_padding_ties_dtypes(golden_dir, oracle):
    g = np.load(golden_dir / "edge_cases.npz")
    X, Q = g["X"], g["Q"]
    for metric in ("l2", "cosine", "ip"):
        for k in (3, 7):
            Dg, Ig = g[f"D_{metric}_k{k}"], g[f"I_{metric}_k{k}"]
            assert Dg.dtype == np.float32 and Ig.dtype == np.int64
            # literal numpy restatement is identical (same argpartition/argsort)
            d_np, i_np = rs.linear_searcher_batch(X, Q, k, metric)
            np.testing.assert_array_equal(i_np, Ig)
            np.testing.assert_array_equal(d_np, Dg)
            # canonical oracle: same neighbours up to the order inside exact ties; same padding
            Xn, Qn, m = _norm_for(metric, X, Q)
            d_c, i_c = oracle.knn(Xn, Qn, k, m)
            d_c, i_c = rs.flat_to_linear(d_c, i_c, m)
            if k == 7:
                assert np.all(i_c[:, 5:] == -1) and np.all(np.isinf(d_c[:, 5:]))
                assert np.all(Ig[:, 5:] == -1) and np.all(np.isinf(Dg[:, 5:]))
                assert_same_neighbours_modulo_ties(i_c[:, :5], Ig[:, :5], d_c[:, :5], Dg[:, :5])
            else:
                np.testing.assert_allclose(d_c, Dg, rtol=1e-6, atol=1e-7)
    # the canonical tie rule: (distance, smaller id)
    d_c, i_c = oracle.knn(X, Q, 5, "l2")
    assert i_c[1].tolist() == [0, 4, 1, 2, 3]
    assert i_c[0].tolist() == [0, 4, 1, 2, 3]
    # zero-norm rows under cosine score -0.0, never NaN
    assert not np.isnan(g["D_cosine_k3"]).any()
    # 1-D search() output and float64 / Fortran inputs
    assert g["D_search1d"].shape == (3,) and g["I_search1d"].shape == (3,)
    X1, Q1 = _kat1()
    d, i = oracle.knn(np.asfortranarray(X1[:50].astype(np.float64)), Q1.astype(np.float64), 4, "l2")
    np.testing.assert_array_equal(i, g["I_f64F"])
    np.testing.assert_allclose(np.sqrt(d), g["D_f64F"], rtol=1e-6)


def test_recall_at_k_known_answers(golden_dir):
    man = json.loads((golden_dir / "manifest.json").read_text())["recall_at_k"]
    gt, pr = np.array(man["gt"]), np.array(man["pred"])
    for key, k in (("r1", 1), ("r2", 2), ("r4", 4), ("r10", 10)):
        assert rs.recall_at_k(gt, pr, k) == pytest.approx(man[key])
    g2 = np.load(golden_dir / "kat2_random_10000x128.npz")
    assert rs.recall_at_k(g2["GT"], g2["I_l2"], 10) == 1.0
    X, Q = _inputs("kat2")
    np.testing.assert_array_equal(rs.ground_truth_l2(X, Q[:5], 100), g2["GT"][:5])


def test_merge_partials_equals_unsharded(oracle):
    X, Q = _kat1()
    for metric in ("l2", "ip"):
        d_all, i_all, k_all = oracle.knn(X, Q, 7, metric, return_keys=True)
        parts = [(0, 400), (400, 650), (650, 1000)]
        keys, ids = [], []
        for lo, hi in parts:
            _, i, kk = oracle.knn(X[lo:hi], Q, 7, metric, id_base=lo, return_keys=True)
            keys.append(kk)
            ids.append(i)
        d_m, i_m = oracle.merge_partials(np.stack(keys), np.stack(ids), metric)
        np.testing.assert_array_equal(i_m, i_all)
        np.testing.assert_array_equal(d_m, d_all)
