#!/usr/bin/env python3
"""IVF-Flat (config 4) searches for rocprofv3 --kernel-trace --stats: nlist=1024 on SIFT1M-shaped data."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np
import vdbhip
from vdbhip import datasets
nprobes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,128").split(",")]
X, Q = datasets.sift_like(1_000_000, 10_000, 128, 1234)
idx = vdbhip.IVFFlatIndex(128, 1024, "l2", 0)
idx.train(X, niter=10); idx.add(X)
for nprobe in nprobes:
    idx.set_nprobe(nprobe)
    for _ in range(6):
        idx.search(Q, 10)
    print(nprobe, idx.stats())
