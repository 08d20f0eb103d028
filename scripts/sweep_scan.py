#!/usr/bin/env python3
"""Interleaved A/B of scan-kernel variants in ONE process (cdna guide rule 24): rounds x variants,
median / min of the HIP-event scan time and of the whole device pipeline."""
import os; os.environ.setdefault('VDBHIP_LIBRARY', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vectordb-retrieval_amd', 'vdbhip', 'libvdbhip_ablations.so'))  # `make -C vectordb-retrieval_amd ablations`
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np
import torch
import vdbhip
sys.path.insert(0, str(ROOT))
from bench import make_data

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sift1m")
ap.add_argument("--variants", default="0,1,2,3,4,5,6")   # "v" or "layout:v" (layout 2 = p16 panels, 0/1 = 32-row tiles)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
X, Q, k, metric = make_data(args.workload, 0)
n, d = X.shape; nq = len(Q)
def parse(v):
    return tuple(int(t) for t in v.split(":")) if ":" in v else (0, int(v))
variants = [parse(v) for v in args.variants.split(",")]
indexes = {}
for lay in sorted({v[0] for v in variants}):
    ix = vdbhip.FlatIndex(d, metric, 0); ix.set_option("panel_layout", lay); ix.add(X); indexes[lay] = ix
dev = torch.device("cuda:0")
q_t = torch.from_numpy(Q).to(dev)
D_t = torch.empty((nq, k), dtype=torch.float32, device=dev); I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
res = {v: {"scan": [], "total": [], "wall": []} for v in variants}
ref = None
for r in range(args.rounds + 1):
    for v in variants:
        idx = indexes[v[0]]
        idx.set_option("scan_variant", v[1])
        idx.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream); torch.cuda.synchronize()
        if ref is None: ref = I_t.clone()
        assert v[1] in (4, 6, 7, 8, 9) or torch.equal(ref, I_t), f"variant {v} changed the result"
        idx.set_option("timing", 1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            idx.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        st = idx.stats(); idx.set_option("timing", 0)
        if r > 0:
            res[v]["scan"].append(st["last_scan_ms"]); res[v]["total"].append(st["last_total_ms"]); res[v]["wall"].append(wall)
flops = 2.0 * nq * n * d
for v in variants:
    s = np.array(res[v]["scan"]); t = np.array(res[v]["total"]); w = np.array(res[v]["wall"])
    print(json.dumps({"layout": v[0], "variant": v[1], "scan_ms_med": round(float(np.median(s)), 4), "scan_ms_min": round(float(s.min()), 4),
                      "pipeline_ms_med": round(float(np.median(t)), 4), "wall_ms_med": round(float(np.median(w)), 4),
                      "TFLOPs_med": round(flops / np.median(s) / 1e9, 1)}))
