#!/usr/bin/env python3
"""IVF-Flat (config 4 shape) A/B of the waves-per-work-item choice (option ivf_nw: 0 auto, 2, 4, 8; every setting is
exact) in ONE process: per nprobe, median device time of the whole search and of the list scan, ids compared."""
import argparse, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np
import torch
import vdbhip
from bench import make_data

ap = argparse.ArgumentParser()
ap.add_argument("--nprobes", default="8,32,128")
ap.add_argument("--nws", default="0,2,4,8")
ap.add_argument("--option", default="ivf_nw")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--workload", default="sift1m", help="sift1m (nlist 1024) | msmarco_ivf (100 000 x 384, nlist 100, k = 20)")
ap.add_argument("--nlist", type=int, default=0)
args = ap.parse_args()
X, Q, k, metric = make_data(args.workload, 0)
dev = torch.device("cuda:0")
idx = vdbhip.IVFFlatIndex(X.shape[1], args.nlist or (100 if args.workload == "msmarco_ivf" else 1024), metric, 0)
idx.train(X)
idx.add(X)
q_t = torch.from_numpy(Q).to(dev)
D_t = torch.empty((len(Q), k), dtype=torch.float32, device=dev)
I_t = torch.empty((len(Q), k), dtype=torch.int64, device=dev)
for nprobe in [int(v) for v in args.nprobes.split(",")]:
    idx.set_nprobe(nprobe)
    nws = [int(v) for v in args.nws.split(",")]
    res = {v: {"scan": [], "total": []} for v in nws}
    ref = None
    for r in range(args.rounds + 1):
        for v in nws:
            idx.set_option(args.option, v)
            for _ in range(2):
                idx.search_device(q_t.data_ptr(), len(Q), k, D_t.data_ptr(), I_t.data_ptr())
            torch.cuda.synchronize()
            idx.set_option("timing", 1)
            for _ in range(args.steps):
                idx.search_device(q_t.data_ptr(), len(Q), k, D_t.data_ptr(), I_t.data_ptr())
            torch.cuda.synchronize()
            st = idx.stats()
            idx.set_option("timing", 0)
            if r == 0:
                ids = I_t.cpu().numpy()
                ref = ids if ref is None else ref
                assert np.array_equal(ids, ref), f"ivf_nw {v} disagrees"
                continue
            res[v]["scan"].append(st["last_scan_ms"])
            res[v]["total"].append(st["last_total_ms"])
    for v in nws:
        print(json.dumps({"nprobe": nprobe, args.option: v, "scan_ms_med": round(float(np.median(res[v]["scan"])), 4),
                          "total_ms_med": round(float(np.median(res[v]["total"])), 4),
                          "candidates_per_query": round(st["last_candidates"] / len(Q), 2), "rescan_bins": st["last_rescan_bins"]}))
