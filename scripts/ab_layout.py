import sys, json, time
from pathlib import Path
ROOT = Path("/root/repo") if Path("/root/repo/bench.py").exists() else Path.cwd()
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from bench import make_data
wl = sys.argv[1] if len(sys.argv) > 1 else "sift1m"
X, Q, k, metric = make_data(wl, 0)
n, d = X.shape; nq = len(Q)
dev = torch.device("cuda:0")
q_t = torch.from_numpy(Q).to(dev)
D_t = torch.empty((nq, k), dtype=torch.float32, device=dev); I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
cfgs = [(0, 0), (2, 0), (2, 16), (2, 32)]
idx = {}
for lay in (0, 2):
    ix = vdbhip.FlatIndex(d, metric, 0); ix.set_option("panel_layout", lay); ix.add(X); idx[lay] = ix
res = {c: [] for c in cfgs}; ref = None
for r in range(6):
    for c in cfgs:
        ix = idx[c[0]]; ix.set_option("spans_per_chunk", c[1])
        ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream); torch.cuda.synchronize()
        if ref is None: ref = I_t.clone()
        assert torch.equal(ref, I_t), c
        ix.set_option("timing", 1)
        for _ in range(5): ix.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
        torch.cuda.synchronize(); st = ix.stats(); ix.set_option("timing", 0)
        if r: res[c].append((st["last_scan_ms"], st["last_total_ms"], st["last_candidates"] / nq, st["last_rescan_bins"]))
for c in cfgs:
    a = np.array(res[c]); print(json.dumps({"layout": c[0], "spc": c[1], "scan_ms": round(float(np.median(a[:, 0])), 4), "pipeline_ms": round(float(np.median(a[:, 1])), 4), "cand_per_q": round(float(a[0, 2]), 2), "rescan_bins": int(a[0, 3])}))
