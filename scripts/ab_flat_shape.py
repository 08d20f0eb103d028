#!/usr/bin/env python3
"""A/B of the two MFMA shapes of the flat scans, D <= 128 (option flat_shape, fixed at vdb_add: two indexes over the same corpus:
32 = 32x32x16 f16 / 32x32x32 i8, 16 = 16x16x32 f16 / 16x16x64 i8 on layout "x16") in ONE process, interleaved rounds, results
compared bit for bit:
    python scripts/ab_flat_shape.py [workload=sift1m] [rounds=5] [nq=0 (the workload's) | n1,n2,.. (several batch sizes)]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from bench import make_data
wl = sys.argv[1] if len(sys.argv) > 1 else "sift1m"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
X, Q, k, metric = make_data(wl, 0)
nqs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3] != "0" else [len(Q)]
dev = torch.device("cuda:0")
ixs = {}
for shape in (32, 16):
    ix = vdbhip.FlatIndex(X.shape[1], metric, 0); ix.set_option("flat_shape", shape); ix.add(X); ixs[shape] = ix
Qall = Q
for nq in nqs:
    Q = Qall[:nq]
    q_t = torch.from_numpy(Q.copy()).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev); I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    st_ = torch.cuda.current_stream().cuda_stream
    res = {s: [] for s in ixs}; ref = None
    for r in range(rounds + 1):
        for shape, ix in ixs.items():
            for _ in range(2): ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), st_)
            torch.cuda.synchronize()
            if ref is None: ref = (I_t.clone(), D_t.clone())
            assert torch.equal(ref[0], I_t) and torch.equal(ref[1], D_t), shape
            ix.set_option("timing", 1)
            for _ in range(6): ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), st_)
            torch.cuda.synchronize(); s = ix.stats(); ix.set_option("timing", 0)
            assert s["scan_shape"] == shape and s["last_path_name"] == "mfma_scan", s
            if r: res[shape].append((s["last_scan_ms"], s["last_total_ms"], s["last_tail_ms"]))
    for shape in ixs:
        a = np.array(res[shape])
        print(json.dumps({"workload": wl, "nq": nq, "flat_shape": shape, "scan_dtype": s["scan_dtype"], "scan_ms": round(float(np.median(a[:, 0])), 4),
                          "pipeline_ms": round(float(np.median(a[:, 1])), 4), "tail_ms": round(float(np.median(a[:, 2])), 4),
                          "qps": round(nq / float(np.median(a[:, 1])) * 1e3)}))
