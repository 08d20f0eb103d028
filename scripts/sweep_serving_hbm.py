#!/usr/bin/env python3
"""Single-query scan of a byte-valued corpus larger than the Infinity Cache (4M x 128 = 512 MB of int8 panels, generated on
the device): A/B of one option in ONE process -- median HIP-event time of the scan over 30 searches per setting, rounds
interleaved, ids compared.  Usage: python scripts/sweep_serving_hbm.py [--option i8_nt] [--values 1,2] [--nq 1]"""
import argparse, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np
import torch
import vdbhip
from bench import device_byte_rows

ap = argparse.ArgumentParser()
ap.add_argument("--option", default="i8_nt")
ap.add_argument("--values", default="1,2")
ap.add_argument("--nq", type=int, default=1)
ap.add_argument("--rows", type=int, default=4_000_000)
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
X = device_byte_rows(a.rows, 128, dev, 77)
idx = vdbhip.FlatIndex(128, "l2", 0)
idx.add_device(X.data_ptr(), a.rows, id_base=0)
del X
torch.cuda.empty_cache()
q = device_byte_rows(64, 128, dev, 78)
D = torch.empty((64, 10), dtype=torch.float32, device=dev)
I = torch.empty((64, 10), dtype=torch.int64, device=dev)
vals = [float(v) for v in a.values.split(",")]
res = {v: [] for v in vals}
ref = None
for r in range(a.rounds + 1):
    for v in vals:
        idx.set_option(a.option, v)
        for _ in range(3):
            idx.search_device(q.data_ptr(), a.nq, 10, D.data_ptr(), I.data_ptr())
        torch.cuda.synchronize()
        idx.set_option("timing", 1)
        for _ in range(30):
            idx.search_device(q.data_ptr(), a.nq, 10, D.data_ptr(), I.data_ptr())
        torch.cuda.synchronize()
        st = idx.stats()
        idx.set_option("timing", 0)
        ids = I[:a.nq].cpu().numpy().copy()
        ref = ids if ref is None else ref
        assert np.array_equal(ids, ref), f"{a.option}={v} disagrees"
        if r:
            res[v].append((st["last_scan_ms"], st["last_total_ms"]))
for v in vals:
    scan = float(np.median([x[0] for x in res[v]])) * 1e3
    total = float(np.median([x[1] for x in res[v]])) * 1e3
    print(json.dumps({a.option: v, "nq": a.nq, "scan_us": round(scan, 1), "pipeline_us": round(total, 1),
                      "scan_tb_s": round(a.rows * 128 / scan / 1e6, 2)}))
