#!/usr/bin/env python3
"""A/B of one vdb_set_option value on a flat bench workload in ONE process (interleaved rounds, ids compared):
    python scripts/sweep_flat_option.py <workload> <option> <v1,v2,...> [rounds]
prints per value the median scan / pipeline / tail time from the library's HIP events and the candidate statistics."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from bench import make_data
wl, opt, vals = sys.argv[1], sys.argv[2], [float(v) for v in sys.argv[3].split(",")]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
X, Q, k, metric = make_data(wl, 0)
nq = len(Q); dev = torch.device("cuda:0")
ix = vdbhip.FlatIndex(X.shape[1], metric, 0); ix.add(X)
q_t = torch.from_numpy(Q).to(dev)
D_t = torch.empty((nq, k), dtype=torch.float32, device=dev); I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
st_ = torch.cuda.current_stream().cuda_stream
res = {v: [] for v in vals}; ref = None
for r in range(rounds + 1):
    for v in vals:
        ix.set_option(opt, v)
        for _ in range(2): ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), st_)
        torch.cuda.synchronize()
        if ref is None: ref = I_t.clone()
        assert torch.equal(ref, I_t), (opt, v)
        ix.set_option("timing", 1)
        for _ in range(6): ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), st_)
        torch.cuda.synchronize(); s = ix.stats(); ix.set_option("timing", 0)
        if r: res[v].append((s["last_scan_ms"], s["last_total_ms"], s["last_tail_ms"], s["last_candidates"] / nq, s["last_rescan_bins"]))
for v in vals:
    a = np.array(res[v])
    print(json.dumps({"workload": wl, opt: v, "scan_ms": round(float(np.median(a[:, 0])), 4), "pipeline_ms": round(float(np.median(a[:, 1])), 4),
                      "tail_ms": round(float(np.median(a[:, 2])), 4), "groups_per_query": round(float(a[0, 3]), 2), "rescan_bins": int(a[0, 4]),
                      "qps": round(nq / float(np.median(a[:, 1])) * 1e3)}))
