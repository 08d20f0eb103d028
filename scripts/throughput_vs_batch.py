#!/usr/bin/env python3
"""Throughput against batch size on the 1M x 128 flat index (device-resident queries, k = 10): median wall time of one
vdb_search_device + synchronise, with the small-batch grid shapes (default) and with the batch-shaped grid for every
size (option small_batch = 0).  Usage: python scripts/throughput_vs_batch.py [--kind sift|gaussian]"""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from vdbhip import datasets

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="sift")
a = ap.parse_args()
if a.kind == "sift":
    X, Q = datasets.sift_like(1_000_000, 10_000, 128, 1234)
else:
    rng = np.random.default_rng(5)
    X, Q = rng.standard_normal((1_000_000, 128), dtype=np.float32), rng.standard_normal((10_000, 128), dtype=np.float32)
idx = vdbhip.FlatIndex(128, "l2", 0)
idx.add(X)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
q_t = torch.from_numpy(Q).to(dev)
D_t = torch.empty((10_000, 10), dtype=torch.float32, device=dev)
I_t = torch.empty((10_000, 10), dtype=torch.int64, device=dev)
for nq in (1, 16, 64, 128, 256, 512, 1024, 2048, 2049, 4096, 10_000):
    row = {"kind": a.kind, "nq": nq}
    for sb in (1, 0):
        idx.set_option("small_batch", sb)
        ts = []
        for it in range(25):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            idx.search_device(q_t.data_ptr(), nq, 10, D_t.data_ptr(), I_t.data_ptr(), stream)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts[5:]))
        row["us" if sb else "us_batch_shape"] = round(med * 1e6, 1)
        row["qps" if sb else "qps_batch_shape"] = round(nq / med)
    print(json.dumps(row), flush=True)
