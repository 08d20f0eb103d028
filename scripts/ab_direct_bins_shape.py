import sys, time, json
from pathlib import Path
ROOT = Path("/root/repo")
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
rng = np.random.default_rng(0)
Q = rng.standard_normal((10000, 128)).astype(np.float32)
dev = torch.device("cuda:0")
q_t = torch.from_numpy(Q).to(dev)
for n in (32768, 50000, 100000, 200000):
    X = rng.standard_normal((n, 128)).astype(np.float32)
    row = {"n": n}
    for shape in (32, 16):
        idx = vdbhip.FlatIndex(128, "l2", 0); idx.set_option("flat_shape", shape); idx.add(X)
        for k in (10, 100):
            D_t = torch.empty((10000, k), dtype=torch.float32, device=dev); I_t = torch.empty((10000, k), dtype=torch.int64, device=dev)
            for _ in range(3): idx.search_device(q_t.data_ptr(), 10000, k, D_t.data_ptr(), I_t.data_ptr())
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); idx.search_device(q_t.data_ptr(), 10000, k, D_t.data_ptr(), I_t.data_ptr()); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            st = idx.stats()
            row[f"shape{shape}_k{k}"] = f"{np.median(ts) * 1e3:.3f}ms/{st['last_path_name'][:5]}/cand{st['last_candidates'] / 10000:.1f}"
        idx.close()
    print(json.dumps(row), flush=True)
