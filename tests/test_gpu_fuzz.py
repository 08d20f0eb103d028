"""Randomised parity sweep: seeded shapes / value distributions across every search path of the flat index (exhaustive
exact, dense small-corpus, MFMA scan, K-loop scan) against the CPU oracle -- ids and distances bit for bit.
Deterministic (fixed seeds); sized so the oracle side finishes in seconds."""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _values(rng, kind, shape):
    if kind == "gauss":
        return rng.standard_normal(shape)
    if kind == "ints":            # SIFT-like: exact in fp16, unscaled
        return np.clip(np.rint(rng.gamma(0.6, 40.0, size=shape)), 0, 218)
    if kind == "bigints":         # integers beyond the unscaled fp16 range
        return np.rint(rng.standard_normal(shape) * 3000.0)
    if kind == "tiny":
        return rng.standard_normal(shape) * 1e-4
    if kind == "huge":
        return rng.standard_normal(shape) * 1e5
    if kind == "heavy":           # heavy tails: a few coordinates dominate the norms
        return np.clip(rng.standard_cauchy(shape), -1e3, 1e3)
    if kind == "sparse":          # mostly zeros
        return rng.standard_normal(shape) * (rng.random(shape) < 0.05)
    if kind == "offset":          # large common offset, small spread (cancellation in ||x||^2 - 2 q.x)
        return 50.0 + rng.standard_normal(shape) * 0.1
    if kind == "bytes":           # the whole uint8 range: int8 scan copy, largest accumulator magnitudes
        return rng.integers(0, 256, size=shape).astype(np.float64)
    if kind == "sbytes":          # the whole int8 range (s8 window)
        return rng.integers(-128, 128, size=shape).astype(np.float64)
    raise AssertionError(kind)


KINDS = ["gauss", "ints", "bigints", "tiny", "huge", "heavy", "sparse", "offset", "bytes", "sbytes"]


def _cases():
    rng = np.random.default_rng(20240607)
    out = []
    # (n range, d range, nq range) per targeted path
    targets = [
        ((1, 3000), (1, 260), (1, 40)),            # exhaustive exact kernel
        ((600, 8192), (2, 128), (64, 200)),        # dense small-corpus MFMA path
        ((32768, 70000), (3, 128), (1, 200)),      # flat MFMA scan
        ((32768, 50000), (129, 300), (1, 120)),    # K-loop MFMA scan (p16 panels)
    ]
    import os
    per_target = int(os.environ.get("VDBHIP_FUZZ_CASES", "10"))     # (a one-off deep sweep: VDBHIP_FUZZ_CASES=100)
    for t, (nr, dr, qr) in enumerate(targets):
        for i in range(per_target):
            n = int(rng.integers(nr[0], nr[1] + 1))
            d = int(rng.integers(dr[0], dr[1] + 1))
            nq = int(rng.integers(qr[0], qr[1] + 1))
            k = int(min(n + 3, rng.choice([1, 2, 5, 10, 33, 100])))
            out.append((t, i, n, d, nq, k, str(rng.choice(["l2", "ip"])), str(rng.choice(KINDS))))
    return out


@pytest.mark.parametrize("target,i,n,d,nq,k,metric,kind", _cases())
def test_random_shapes_and_distributions_bit_exact(oracle, target, i, n, d, nq, k, metric, kind):
    import vdbhip

    rng = np.random.default_rng(1000 * target + i)
    X = _values(rng, kind, (n, d)).astype(np.float32)
    Q = _values(rng, kind, (nq, d)).astype(np.float32)
    if n > 10 and i % 3 == 0:      # duplicates, zero rows and a query that IS a corpus row
        X[n // 2] = X[n // 3]
        X[n // 5] = 0.0
        Q[0] = X[n // 7]
    # round 4: every third case runs on another form of the engine (same results whatever the form): several shards in one
    # handle (vdb_create_multi), an int8-only index (byte-valued corpora of > 32 768 rows take it, the rest fall back to the
    # default layout), quads instead of octs as the fp16 candidate group, the 32x32x32 shape of the int8 scan
    form = (target * 7 + i) % 6
    idx = vdbhip.FlatIndex(d, metric, [0, 0, 0] if form == 1 else [0, 0] if form == 4 else 0)
    if form in (2, 4):
        idx.set_option("int8_only", 1)
    if form == 3:
        idx.set_option("f16_group", 4)
    if form == 5 or (form == 2 and i % 2):
        idx.set_option("i8_shape", 32)      # the 32x32x32 int8 scan and its panel layout (default: 16x16x64, layout "x16")
    idx.add(X, id_base=11)
    D, I = idx.search(Q, k)
    Do, Io = oracle.knn(X, Q, k, metric, id_base=11)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    shards = 2 if form == 4 else 1       # (the int8-only layout is per shard: more than 32 768 rows each)
    if form in (2, 4) and kind in ("bytes", "sbytes", "ints") and -(-n // shards) > 32768 and n % shards == 0 and d <= 128:
        assert idx.stats()["has_i8_copy"] == 2
        Qf = (Q + 0.37).astype(np.float32)          # a non-integer batch on the int8-only index: converted fp16 slabs
        D, I = idx.search(Qf, k)
        Do, Io = oracle.knn(X, Qf, k, metric, id_base=11)
        np.testing.assert_array_equal(I, Io)
        np.testing.assert_array_equal(D, Do)
    idx.close()


def _ivf_cases():
    import os
    rng = np.random.default_rng(77)
    out = []
    for i in range(int(os.environ.get("VDBHIP_FUZZ_IVF_CASES", "12"))):
        n = int(rng.integers(2000, 70000))
        d = int(rng.integers(2, 200))
        nlist = int(rng.integers(2, min(400, n // 20)))
        nq = int(rng.integers(1, 400))
        k = int(rng.choice([1, 5, 10, 50]))
        out.append((i, n, d, nlist, int(rng.integers(1, nlist + 1)), nq, k, str(rng.choice(["l2", "ip"])),
                    str(rng.choice(["gauss", "ints", "heavy", "offset", "sparse"]))))
    return out


def _ivf_cases_wide():
    """Round 3: the list-major scans that are new or re-laid -- D > 128 (items-mode K-loop scan, both span sizes), and
    byte-valued corpora with D <= 128 (int8 items-mode scan: 256-row spans, run-ordered vector bin stores, staging ring)."""
    import os
    rng = np.random.default_rng(303)
    out = []
    for i in range(int(os.environ.get("VDBHIP_FUZZ_IVF_WIDE_CASES", "12"))):
        wide = i % 2 == 0
        n = int(rng.integers(15000, 60000))
        d = int(rng.integers(129, 800)) if wide else int(rng.integers(8, 129))
        nlist = int(rng.integers(4, 80))
        nq = int(rng.integers(1, 700))
        k = int(rng.choice([1, 5, 10, 20, 50]))
        kind = str(rng.choice(["gauss", "ints", "heavy", "offset"] if wide else ["bytes", "sbytes", "ints"]))
        out.append((100 + i, n, d, nlist, int(rng.integers(1, nlist + 1)), nq, k, str(rng.choice(["l2", "ip"])), kind))
    return out


@pytest.mark.parametrize("i,n,d,nlist,nprobe,nq,k,metric,kind", _ivf_cases() + _ivf_cases_wide())
def test_random_ivf_configurations_bit_exact(oracle, i, n, d, nlist, nprobe, nq, k, metric, kind):
    """IVF-Flat with injected centroids == brute force restricted to the probed lists (oracle/ivf_oracle.c), for random
    list counts (including empty and tiny lists), probe counts, dims on both sides of the MFMA list-scan limits."""
    import vdbhip

    rng = np.random.default_rng(500 + i)
    X = _values(rng, kind, (n, d)).astype(np.float32)
    Q = _values(rng, kind, (nq, d)).astype(np.float32)
    C = X[rng.choice(n, nlist, replace=False)].copy()
    if i % 4 == 0:
        C[0] = 1e4            # a centroid far from everything: an empty list
    idx = vdbhip.IVFFlatIndex(d, nlist, metric, [0, 0] if i % 5 == 1 else 0)       # (round 4: some cases over two shards)
    idx.set_centroids(C)
    if d > 128 and i % 3 == 0:
        idx.set_option("ivf_tps", 64)       # 1024-row spans / 256-row bins instead of the automatic choice
    if d > 128 and i % 4 == 2:
        idx.set_option("ivf_group", [4, 2][(i // 4) % 2])      # quads / pairs instead of single rows as the candidate group
    if d <= 128 and i % 4 == 3:
        idx.set_option("ivf_i8_group", 8)
    if d > 128 and i % 2 == 1:
        idx.set_option("ivf_tile", 2)           # the square workgroup tile of the K-loop list scan (hand-over of minima through LDS)
    idx.add(X, id_base=3)
    lor = idx.assignment()
    np.testing.assert_array_equal(lor, oracle.ivf_assign(C, X, metric))
    idx.set_nprobe(nprobe)
    D, I = idx.search(Q, k)
    Do, Io = oracle.ivf_search(X, C, lor, Q, k, nprobe, metric, id_base=3)
    np.testing.assert_array_equal(I, Io)
    np.testing.assert_array_equal(D, Do)
    idx.close()
