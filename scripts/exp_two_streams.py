#!/usr/bin/env python3
"""Experiment: does overlapping the tail of one half-batch with the scan of the other pay?  Two handles over the same
1M x 128 corpus, the 10k-query batch split in two halves on two streams (the first with high priority), against one
10k-query call.  Usage: python scripts/exp_two_streams.py"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from bench import make_data

X, Q, k, metric = make_data("sift1m", 0)
dev = torch.device("cuda", 0)
ia, ib = vdbhip.FlatIndex(128, metric, 0), vdbhip.FlatIndex(128, metric, 0)
ia.add(X); ib.add(X)
q_t = torch.from_numpy(Q).to(dev)
D_t = torch.empty((len(Q), k), dtype=torch.float32, device=dev)
I_t = torch.empty((len(Q), k), dtype=torch.int64, device=dev)
s_hi, s_lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
s_one = torch.cuda.Stream()

def one():
    ia.search_device(q_t.data_ptr(), len(Q), k, D_t.data_ptr(), I_t.data_ptr(), s_one.cuda_stream)

def two(split):
    ia.search_device(q_t.data_ptr(), split, k, D_t.data_ptr(), I_t.data_ptr(), s_hi.cuda_stream)
    ib.search_device(q_t[split:].data_ptr(), len(Q) - split, k, D_t[split:].data_ptr(), I_t[split:].data_ptr(), s_lo.cuda_stream)

def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e6, 1)

one(); torch.cuda.synchronize(); ref = I_t.clone()
print(json.dumps({"one_call_us": timeit(one)}))
for split in (5120, 4096, 6144):
    two(split); torch.cuda.synchronize()
    assert torch.equal(ref, I_t)
    print(json.dumps({"split": split, "two_streams_us": timeit(lambda: two(split))}))
print(json.dumps({"one_call_us_again": timeit(one)}))
