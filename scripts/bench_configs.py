#!/usr/bin/env python3
"""BASELINE.json configs 1-4 on one MI355X: recall + QPS through the PLUGIN API (host numpy in/out, i.e. the
reference harness's view: qps = n_queries / wall(batch_search), experiment_runner.py:431-464) and through the
device-resident API.  Writes one JSON line per measurement (profiles/r01_configs.jsonl)."""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np
import torch
import vdbhip
from vdbhip import datasets
from vdbhip.metrics import recall_at_k

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="1,2,2g,3,4")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "configs.jsonl"))
args = ap.parse_args()
out = open(args.out, "a")


def emit(**kw):
    line = json.dumps(kw)
    print(line, flush=True)
    out.write(line + "\n"); out.flush()


def timed(fn, reps):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return r, float(np.median(ts)), float(ts[0])


def device_qps(index, Q, k, reps):
    dev = torch.device("cuda:0")
    q = torch.from_numpy(Q).to(dev); nq = len(Q)
    D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        index.search_device(q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), st); torch.cuda.synchronize()
    _, med, _ = timed(run, reps)
    return nq / med, I.cpu().numpy()


def flat_case(tag, X, Q, k, metric, reps, gt=None):
    algo = vdbhip.get_algorithm_instance("HipExactSearch", X.shape[1], name=tag, metric=metric)
    t0 = time.perf_counter(); algo.build_index(X); build = time.perf_counter() - t0
    (D, I), med, first = timed(lambda: algo.batch_search(Q, k=k), reps)
    dq, I2 = device_qps(algo.index, Q, k, reps)
    assert np.array_equal(I, I2)
    st = algo.index.stats()
    emit(config=tag, n=len(X), dim=X.shape[1], nq=len(Q), k=k, metric=metric, path=st["last_path_name"],
         qps_plugin_host_io=round(len(Q) / med, 1), qps_device_resident=round(dq, 1), first_call_s=round(first, 5),
         build_s=round(build, 3), candidates_per_query=round(st["last_candidates"] / len(Q), 2),
         rescan_bins=st["last_rescan_bins"], fallback_queries=st["last_fallback_queries"],
         recall_at_10=None if gt is None else recall_at_k(gt, I, 10), hbm_mb=round(st["bytes_resident"] / 2**20, 1))
    return algo, I


cfgs = args.configs.split(",")
if "1" in cfgs:   # plumbing config: reference generator, GT by the reference's recipe
    X, Q = datasets.random_reference(128, 10000, 100, 42)
    gt = np.stack([np.argsort(np.linalg.norm(X - q, axis=1))[:100] for q in Q]).astype(np.int32)
    flat_case("config1_random_10000x128", X, Q, 10, "l2", args.reps, gt)
if "2" in cfgs or "4" in cfgs:
    Xs, Qs = datasets.sift_like(1_000_000, 10_000, 128, 1234)
if "2" in cfgs:
    algo, I_exact = flat_case("config2_sift1m_like", Xs, Qs, 10, "l2", args.reps)
    del algo
if "2g" in cfgs:
    Xg, Qg = datasets.gaussian(1_000_000, 10_000, 128, 1234)
    a, _ = flat_case("config2_gaussian1m", Xg, Qg, 10, "l2", args.reps); del a, Xg, Qg
if "3" in cfgs:
    Xv, Qv = datasets.glove_like(1_200_000, 10_000, 50, 50)
    a, _ = flat_case("config3_glove50_like_ip", Xv, Qv, 10, "ip", args.reps); del a
    c = vdbhip.CompositeAlgorithm(name="cos", dimension=50, metric="cosine",
                                  indexer={"type": "HipBruteForceIndexer", "metric": "cosine"},
                                  searcher={"type": "HipLinearSearcher", "metric": "cosine"})
    c.build_index(Xv)
    (_, Ic), med, first = timed(lambda: c.batch_search(Qv, k=10), args.reps)
    emit(config="config3_glove50_like_cosine", n=len(Xv), dim=50, nq=len(Qv), k=10, metric="cosine",
         qps_plugin_host_io=round(len(Qv) / med, 1), first_call_s=round(first, 5))
    del c, Xv, Qv
if "4" in cfgs:   # IVF-Flat nlist=1024 on the config-2 data; GT = exact result
    if "2" not in cfgs:
        e = vdbhip.FlatIndex(128, "l2", 0); e.add(Xs); _, I_exact = e.search(Qs, 10); e.close()
    idx = vdbhip.IVFFlatIndex(128, 1024, "l2", 0)
    t0 = time.perf_counter(); idx.train(Xs, niter=25, seed=1234); train = time.perf_counter() - t0
    t0 = time.perf_counter(); idx.add(Xs); add = time.perf_counter() - t0
    sizes = np.bincount(idx.assignment(), minlength=1024)
    for nprobe in (1, 8, 32, 128):
        idx.set_nprobe(nprobe)
        (D, I), med, first = timed(lambda: idx.search(Qs, 10), max(3, args.reps // 2))
        dq, _ = device_qps(idx, Qs, 10, max(3, args.reps // 2))
        emit(config="config4_ivf1024_flat_sift1m_like", nprobe=nprobe, nq=len(Qs), k=10,
             recall_at_10_vs_exact=round(recall_at_k(I_exact, I, 10), 5), qps_plugin_host_io=round(len(Qs) / med, 1),
             qps_device_resident=round(dq, 1), train_s=round(train, 2), add_s=round(add, 2),
             list_min=int(sizes.min()), list_max=int(sizes.max()))
