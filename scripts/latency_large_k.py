#!/usr/bin/env python3
"""Large-k searches of the ground-truth builder (k = 100 / 200 on 1M rows): device time per 10k-query
batch of the flat index, SIFT1M-shaped (int8 scan) and Gaussian (fp16 scan)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np
import torch
import vdbhip
from bench import make_data
dev = torch.device("cuda:0")
for wl in ("sift1m", "gaussian1m"):
    X, Q, _, metric = make_data(wl, 0)
    idx = vdbhip.FlatIndex(X.shape[1], metric, 0)
    idx.add(X)
    q_t = torch.from_numpy(Q).to(dev)
    for k in (10, 100, 200):
        D_t = torch.empty((len(Q), k), dtype=torch.float32, device=dev)
        I_t = torch.empty((len(Q), k), dtype=torch.int64, device=dev)
        for _ in range(2):
            idx.search_device(q_t.data_ptr(), len(Q), k, D_t.data_ptr(), I_t.data_ptr())
        torch.cuda.synchronize()
        idx.set_option("timing", 1)
        for _ in range(5):
            idx.search_device(q_t.data_ptr(), len(Q), k, D_t.data_ptr(), I_t.data_ptr())
        torch.cuda.synchronize()
        st = idx.stats()
        idx.set_option("timing", 0)
        print(json.dumps({"workload": wl, "k": k, "scan_ms": round(st["last_scan_ms"], 3), "total_ms": round(st["last_total_ms"], 3),
                          "path": st["last_path_name"], "dtype": st["scan_dtype"], "cand_per_q": round(st["last_candidates"] / len(Q), 1),
                          "rescans": st["last_rescan_bins"], "fallbacks": st["last_fallback_queries"]}))
    idx.close()
