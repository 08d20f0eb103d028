#!/usr/bin/env python3
"""What the paired launch of the two x16 scans (option scan_pair) and the clear of ws.small inside the prep kernel (option
fused_stats = 0 restores the memset) are worth: wall time of one search_device + stream synchronise, 1M x 128 byte-valued rows,
settings interleaved in ONE process, ids compared.  python scripts/ab_scan_pair.py"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from bench import make_data
X, Q, k, metric = make_data("sift1m", 0)
dev = torch.device("cuda:0")
ix = vdbhip.FlatIndex(X.shape[1], metric, 0); ix.add(X)
settings = {"default": {"scan_pair": 1}, "two launches": {"scan_pair": 0}}
for nq in (1, 16, 64, 200, 10000):
    q_t = torch.from_numpy(Q[:nq].copy()).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev); I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    st_ = torch.cuda.current_stream().cuda_stream
    res = {n: [] for n in settings}; ref = None
    for r in range(6):
        for name, opts in settings.items():
            for o, v in opts.items(): ix.set_option(o, v)
            for _ in range(5): ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), st_)
            torch.cuda.synchronize()
            ts = []
            for _ in range(40 if nq < 1000 else 10):
                t0 = time.perf_counter()
                ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), st_)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ids = I_t.cpu().numpy().copy(); ref = ids if ref is None else ref
            assert np.array_equal(ids, ref), name
            if r: res[name].append(float(np.median(ts)))
    print(json.dumps({"nq": nq, **{n: round(float(np.median(v)) * 1e6, 1) for n, v in res.items()}, "unit": "us wall per search"}))
for o in ("scan_pair",): ix.set_option(o, 1)
