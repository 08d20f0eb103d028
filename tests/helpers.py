"""Shared assertions for parity tests."""
from __future__ import annotations

import numpy as np


def assert_same_neighbours_modulo_ties(ids_a, ids_b, keys_a, keys_b, rtol=0.0):
    """Rows must hold the same ids; inside a group of equal keys any order is accepted
    (the reference's argpartition/argsort leaves the order of exact ties unspecified)."""
    ids_a, ids_b = np.asarray(ids_a), np.asarray(ids_b)
    assert ids_a.shape == ids_b.shape
    for r in range(ids_a.shape[0]):
        if np.array_equal(ids_a[r], ids_b[r]):
            continue
        ka, kb = np.asarray(keys_a[r], np.float64), np.asarray(keys_b[r], np.float64)
        np.testing.assert_allclose(ka, kb, rtol=max(rtol, 1e-6), atol=1e-6)
        # group by key value of row a; the id multisets per group must agree except at the cut
        vals = np.unique(ka)
        for v in vals[:-1]:
            sa = set(ids_a[r][ka == v].tolist())
            sb = set(ids_b[r][np.isclose(kb, v, rtol=max(rtol, 1e-6), atol=1e-6)].tolist())
            assert sa == sb, f"row {r}: tie group {v} differs: {sa} vs {sb}"


def tie_band_mismatch_report(ids_test, ids_ref, keys64_of_test, keys64_of_ref, band=1e-6):
    """For rows whose id lists differ, verify every differing position lies in a near-tie band of the
    exact float64 keys (|ka-kb| <= band*max(1,|k|)).  Returns number of differing rows."""
    bad_rows = 0
    for r in range(ids_test.shape[0]):
        if np.array_equal(ids_test[r], ids_ref[r]):
            continue
        bad_rows += 1
        ka, kb = keys64_of_test[r], keys64_of_ref[r]
        diff = ids_test[r] != ids_ref[r]
        scale = np.maximum(1.0, np.abs(kb[diff]))
        assert np.all(np.abs(ka[diff] - kb[diff]) <= band * scale), (
            f"row {r}: neighbour mismatch outside the tie band: {ka[diff]} vs {kb[diff]}")
    return bad_rows
