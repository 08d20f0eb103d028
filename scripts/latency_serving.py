#!/usr/bin/env python3
"""Small-batch (serving-shaped) latency of the device-resident flat search on SIFT1M-shaped data: wall time of one
`search_device` + stream synchronise for nq = 1 ... 512, median of 30.  Under `rocprofv3 --kernel-trace` with `--nq N`
the trace holds only that batch size (scripts/trace_flat.py gives the per-kernel breakdown).
Usage: python scripts/latency_serving.py [--nq N] [--kind sift|gaussian]"""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from vdbhip import datasets

ap = argparse.ArgumentParser()
ap.add_argument("--nq", type=int, default=0)
ap.add_argument("--kind", default="sift", help="sift | gaussian (1M x 128, l2) | marco (2M x 768, ip, rows generated on device)")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--option", default="", help="name=value[,name=value] options set on the index")
ap.add_argument("--graph", type=int, default=0, help="1: option graph (hipGraph replay of the repeated search)")
ap.add_argument("--ivf", type=int, default=0, help="IVF-Flat nlist = 1024 with this nprobe instead of the flat index")
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.kind == "marco":
    from bench import device_rows
    X_t = device_rows(2_000_000, 768, 0, dev)
    Q = np.random.default_rng(1235).standard_normal((max(512, a.nq), 768), dtype=np.float32)
    idx = vdbhip.FlatIndex(768, "ip", 0)
    idx.add_device(X_t.data_ptr(), 2_000_000, id_base=0)
    torch.cuda.synchronize()
elif a.kind == "sift":
    X, Q = datasets.sift_like(1_000_000, max(512, a.nq), 128, 1234)
else:
    rng = np.random.default_rng(5)
    X, Q = rng.standard_normal((1_000_000, 128), dtype=np.float32), rng.standard_normal((max(512, a.nq), 128), dtype=np.float32)
if a.kind == "marco":
    pass
elif a.ivf:
    idx = vdbhip.IVFFlatIndex(128, 1024, "l2", 0)
    idx.train(X, niter=10, seed=1234, max_points_per_centroid=256)
    idx.add(X)
    idx.set_nprobe(a.ivf)
else:
    idx = vdbhip.FlatIndex(128, "l2", 0)
    idx.add(X)
side = torch.cuda.Stream()          # (a non-null stream: the legacy default stream cannot be captured)
stream = side.cuda_stream
idx.set_option("graph", a.graph)
for kv in filter(None, a.option.split(",")):
    idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
q_t = torch.from_numpy(Q).to(dev)
for nq in ([a.nq] if a.nq else [1, 8, 64, 512]):
    D_t = torch.empty((nq, a.k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, a.k), dtype=torch.int64, device=dev)
    ts = []
    for it in range(40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx.search_device(q_t.data_ptr(), nq, a.k, D_t.data_ptr(), I_t.data_ptr(), stream)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    st = idx.stats()
    print(json.dumps({"kind": a.kind, "ivf_nprobe": a.ivf, "nq": nq, "k": a.k, "median_us": round(float(np.median(ts[10:])) * 1e6, 1),
                      "min_us": round(min(ts[10:]) * 1e6, 1), "graph_replays": st.get("graph_replays", 0), "path": st["last_path_name"],
                      "scan_dtype": int(st.get("scan_dtype", 0))}), flush=True)
