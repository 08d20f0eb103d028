#!/usr/bin/env python3
"""Interleaved A/B of the int8 scan variants (option i8_variant; every variant returns exact results) in ONE process:
rounds x variants, median of the HIP-event scan time and of the whole device pipeline, ids compared with variant 0."""
import argparse, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np
import torch
import vdbhip
from bench import make_data

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="3,6,7,1")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--option", default="i8_variant")
args = ap.parse_args()
X, Q, k, metric = make_data("sift1m", 0)
dev = torch.device("cuda:0")
idx = vdbhip.FlatIndex(X.shape[1], metric, 0)
idx.add(X)
q_t = torch.from_numpy(Q).to(dev)
D_t = torch.empty((len(Q), k), dtype=torch.float32, device=dev)
I_t = torch.empty((len(Q), k), dtype=torch.int64, device=dev)
variants = [int(v) for v in args.variants.split(",")]
res = {v: {"scan": [], "total": []} for v in variants}
ref = None
for r in range(args.rounds + 1):
    for v in variants:
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
            if ref is None:
                ref = ids
            assert np.array_equal(ids, ref), f"variant {v} disagrees with variant {variants[0]}"
            continue
        res[v]["scan"].append(st["last_scan_ms"])
        res[v]["total"].append(st["last_total_ms"])
for v in variants:
    print(json.dumps({args.option: v, "scan_ms_med": round(float(np.median(res[v]["scan"])), 4),
                      "scan_ms_min": round(float(np.min(res[v]["scan"])), 4),
                      "total_ms_med": round(float(np.median(res[v]["total"])), 4), "dtype": st["scan_dtype"]}))
