#!/usr/bin/env python3
"""A/B of the K-loop scan (D > 128) in ONE process: (scan_variant, kloop_qgroup) combinations on a device-
generated Gaussian corpus; every combination must return the ids of the first one."""
import os; os.environ.setdefault('VDBHIP_LIBRARY', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vectordb-retrieval_amd', 'vdbhip', 'libvdbhip_ablations.so'))  # `make -C vectordb-retrieval_amd ablations`
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np
import torch
import vdbhip
from bench import device_rows

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, default=10_000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--metric", default="ip")
ap.add_argument("--combos", default="0:0:0,1:0:0")   # layout:variant:qgroup  (layout 0 = p16 panels, 1 = 32-row tiles)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
X_t = device_rows(args.rows, args.dim, 0, dev)
combos = [tuple(int(x) for x in c.split(":")) for c in args.combos.split(",")]
indexes = {}
for lay in sorted({c[0] for c in combos}):
    ix = vdbhip.FlatIndex(args.dim, args.metric, 0)
    ix.set_option("panel_layout", lay)
    ix.add_device(X_t.data_ptr(), args.rows); torch.cuda.synchronize()
    indexes[lay] = ix
del X_t
nq, k = args.queries, args.k
q_t = torch.from_numpy(np.random.default_rng(1235).standard_normal((nq, args.dim), dtype=np.float32)).to(dev)
D_t = torch.empty((nq, k), dtype=torch.float32, device=dev); I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
res = {c: [] for c in combos}
ref = None
for r in range(args.rounds + 1):
    for c in combos:
        idx = indexes[c[0]]
        idx.set_option("scan_variant", c[1]); idx.set_option("kloop_qgroup", c[2])
        idx.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream); torch.cuda.synchronize()
        if ref is None: ref = I_t.clone()
        assert c[1] >= 7 or torch.equal(ref, I_t), f"combo {c} changed the result"
        idx.set_option("timing", 1)
        for _ in range(args.steps):
            idx.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
        torch.cuda.synchronize()
        st = idx.stats(); idx.set_option("timing", 0)
        if r > 0: res[c].append(st["last_scan_ms"])
flops = 2.0 * nq * args.rows * args.dim
for c in combos:
    s = np.array(res[c])
    print(json.dumps({"layout": "p16" if c[0] == 0 else "p32", "scan_variant": c[1], "kloop_qgroup": c[2], "scan_ms_med": round(float(np.median(s)), 3),
                      "scan_ms_min": round(float(s.min()), 3), "TFLOPs_med": round(flops / np.median(s) / 1e9, 1)}), flush=True)
