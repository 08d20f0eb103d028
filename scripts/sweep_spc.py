#!/usr/bin/env python3
"""Interleaved sweep of the chunk size (spans per workgroup chunk) of the flat scan."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from bench import make_data
X, Q, k, metric = make_data(sys.argv[1] if len(sys.argv) > 1 else "sift1m", 0)
idx = vdbhip.FlatIndex(X.shape[1], metric, 0); idx.add(X)
dev = torch.device("cuda:0"); q_t = torch.from_numpy(Q).to(dev); nq = len(Q)
D_t = torch.empty((nq, k), dtype=torch.float32, device=dev); I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for r in range(4):
    for spc in (8, 12, 16, 24, 32, 48):
        idx.set_option("spans_per_chunk", spc)
        idx.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), st); torch.cuda.synchronize()
        idx.set_option("timing", 1)
        for _ in range(5): idx.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), st)
        torch.cuda.synchronize(); s = idx.stats(); idx.set_option("timing", 0)
        if r: res.setdefault(spc, []).append((s["last_scan_ms"], s["last_total_ms"]))
for spc, v in res.items():
    a = np.array(v); print(json.dumps({"spc": spc, "scan_ms_med": round(float(np.median(a[:, 0])), 4), "pipeline_ms_med": round(float(np.median(a[:, 1])), 4)}))
