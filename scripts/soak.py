#!/usr/bin/env python3
"""Soak: for ~N seconds, random batch sizes against three indices (int8 flat, fp16 flat, K-loop) and an IVF index, device
resident, every result compared bit for bit with the first result of that (index, batch size) -- which itself was checked
against the exact kernels (force_path = 1).  Catches rare races (LDS-DMA / barrier patterns) a single test run cannot.
Usage: python scripts/soak.py [seconds]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
rng = np.random.default_rng(99)
dev = torch.device("cuda", 0)
cases = {}
Xs = np.clip(np.rint(rng.gamma(0.6, 40.0, size=(200_000, 128))), 0, 218).astype(np.float32)
Xg = rng.standard_normal((200_000, 128)).astype(np.float32)
Xk = rng.standard_normal((150_000, 256)).astype(np.float32)
def flat(X, metric):
    ix = vdbhip.FlatIndex(X.shape[1], metric, 0); ix.add(X); return ix
idx = {"i8": flat(Xs, "l2"), "f16": flat(Xg, "l2"), "kloop": flat(Xk, "ip")}
ivf = vdbhip.IVFFlatIndex(128, 256, "l2", 0); ivf.train(Xs, niter=5, seed=1); ivf.add(Xs); ivf.set_nprobe(16); idx["ivf"] = ivf
Q = {"i8": np.clip(np.rint(rng.gamma(0.6, 40.0, size=(3000, 128))), 0, 218).astype(np.float32),
     "f16": rng.standard_normal((3000, 128)).astype(np.float32), "kloop": rng.standard_normal((3000, 256)).astype(np.float32)}
Q["ivf"] = Q["i8"]
qd = {k: torch.from_numpy(v).to(dev) for k, v in Q.items()}
side = torch.cuda.Stream()
sizes = [1, 3, 16, 17, 64, 65, 200, 256, 300, 512, 513, 1024, 2500, 3000]
ref, n, t0 = {}, 0, time.time()
while time.time() - t0 < budget:
    name = list(idx)[int(rng.integers(len(idx)))]
    nq = sizes[int(rng.integers(len(sizes)))]
    ix = idx[name]
    D = torch.empty((nq, 10), dtype=torch.float32, device=dev); I = torch.empty((nq, 10), dtype=torch.int64, device=dev)
    ix.search_device(qd[name].data_ptr(), nq, 10, D.data_ptr(), I.data_ptr(), side.cuda_stream)
    side.synchronize()
    got = (D.cpu().numpy(), I.cpu().numpy())
    key = (name, nq)
    if key not in ref:
        if name != "ivf":
            ix.set_option("force_path", 1)
            De, Ie = ix.search(Q[name][:nq], 10)
            ix.set_option("force_path", 0)
        else:
            ix.set_option("force_path", 1); De, Ie = ix.search(Q[name][:nq], 10); ix.set_option("force_path", 0)
        assert np.array_equal(got[1], Ie) and np.array_equal(got[0], De), f"{key}: differs from the exact kernels"
        ref[key] = got
    else:
        assert np.array_equal(got[1], ref[key][1]) and np.array_equal(got[0], ref[key][0]), f"{key}: run {n} differs from the first"
    n += 1
print(f"soak ok: {n} searches over {len(ref)} (index, batch size) combinations in {time.time() - t0:.0f} s, all identical")
