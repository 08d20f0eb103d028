#!/usr/bin/env python3
"""Wall time of FlatIndex.search (host in/out) over a grid of corpus sizes and batch sizes: looks for cliffs in the
path selection.  D = 128, L2, k = 10 (and k = 100 on the last column)."""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, vdbhip
rng = np.random.default_rng(0)
Qall = rng.standard_normal((10000, 128)).astype(np.float32)
for n in (1000, 5000, 8192, 9000, 20000, 32768, 100000, 1000000):
    X = rng.standard_normal((n, 128)).astype(np.float32)
    idx = vdbhip.FlatIndex(128, "l2", 0); idx.add(X)
    row = {"n": n}
    for nq, k in ((1, 10), (16, 10), (64, 10), (256, 10), (1000, 10), (10000, 10), (10000, 100)):
        q = Qall[:nq]
        idx.search(q, k)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); idx.search(q, k); ts.append(time.perf_counter() - t0)
        row[f"q{nq}_k{k}"] = f"{np.median(ts) * 1e3:.2f}ms/{idx.stats()['last_path_name'][:5]}"
    print(json.dumps(row), flush=True)
    idx.close()
