import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, vdbhip
from vdbhip import datasets
X, Q = datasets.sift_like(1_000_000, 1024, 128, 1234)
idx = vdbhip.FlatIndex(128, "l2", 0); idx.add(X)
for nq in (4, 8, 16, 32, 48, 64, 96, 128, 256, 1024):
    row = {"nq": nq}
    for name, fp in (("blocked", 1), ("single", 3)):
        idx.set_option("force_path", fp)
        q = Q[:nq]
        idx.search(q, 10)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); idx.search(q, 10); ts.append(time.perf_counter() - t0)
        row[name + "_ms"] = round(float(np.median(ts)) * 1e3, 3)
        row[name + "_min"] = round(float(np.min(ts)) * 1e3, 3)
    print(row, flush=True)
