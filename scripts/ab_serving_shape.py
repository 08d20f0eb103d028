#!/usr/bin/env python3
"""Serving-shaped scans (1 / 64 queries) of a byte-valued corpus on both MFMA shapes (option flat_shape = 32 | 16: two indexes over
the same device-generated rows), interleaved in ONE process, ids compared:  python scripts/ab_serving_shape.py [rows=4000000] [f16]
(f16: Gaussian rows and queries -> the fp16 scans)"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from bench import device_byte_rows
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda:0")
f16 = len(sys.argv) > 2 and sys.argv[2] == "f16"
if f16:
    g = torch.Generator(device=dev); g.manual_seed(77)
    X = torch.randn((rows, 128), generator=g, device=dev, dtype=torch.float32)
else:
    X = device_byte_rows(rows, 128, dev, 77)
ixs = {}
for shape in (32, 16):
    ix = vdbhip.FlatIndex(128, "l2", 0); ix.set_option("flat_shape", shape); ix.add_device(X.data_ptr(), rows, id_base=0); ixs[shape] = ix
del X; torch.cuda.empty_cache()
q = torch.randn((64, 128), device=dev, dtype=torch.float32) if f16 else device_byte_rows(64, 128, dev, 78)
D = torch.empty((64, 10), dtype=torch.float32, device=dev); I = torch.empty((64, 10), dtype=torch.int64, device=dev)
for nq in (1, 64):
    res = {s: [] for s in ixs}; ref = None
    for r in range(5):
        for shape, ix in ixs.items():
            for _ in range(3): ix.search_device(q.data_ptr(), nq, 10, D.data_ptr(), I.data_ptr())
            torch.cuda.synchronize()
            ix.set_option("timing", 1)
            for _ in range(30): ix.search_device(q.data_ptr(), nq, 10, D.data_ptr(), I.data_ptr())
            torch.cuda.synchronize(); st = ix.stats(); ix.set_option("timing", 0)
            ids = I[:nq].cpu().numpy().copy(); ref = ids if ref is None else ref
            assert np.array_equal(ids, ref) and st["scan_shape"] == shape and st["scan_dtype"] == (0 if f16 else 1), st
            if r: res[shape].append((st["last_scan_ms"], st["last_total_ms"]))
    for shape in ixs:
        a = np.array(res[shape])
        print(json.dumps({"rows": rows, "nq": nq, "flat_shape": shape, "scan_us": round(float(np.median(a[:, 0])) * 1e3, 1),
                          "pipeline_us": round(float(np.median(a[:, 1])) * 1e3, 1),
                          "dtype": "f16" if f16 else "i8", "scan_tb_s": round(rows * 128 * (2 if f16 else 1) / (float(np.median(a[:, 0])) * 1e-3) / 1e12, 2)}))
