#!/usr/bin/env python3
"""bench.py -- QPS of the exact k-NN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--nprobe P]

A "step" is one pass of the hot path over one 10 000-query batch already resident in HBM:
libvdbhip's device pipeline (query prep -> MFMA scan + bin select -> exact float64 refine), and for
N > 1 the RCCL all-gather of the per-shard partial top-k plus the merge kernel.

N = 1   default workload = BASELINE.json configs[1]: SIFT1M-shaped corpus (1 000 000 x 128, integer-valued float32),
        10 000 queries, k = 10, L2.  Synthetic unless real SIFT1M files are found under $VDBHIP_DATA
        (sift_base.fvecs / sift_query.fvecs / sift_groundtruth.ivecs, read with vdbhip.io's correct reader).
N > 1   default workload = BASELINE.json configs[4] per-GPU shard: 12.5M x 768 inner product, rows generated on
        device (block seeds by global block number).  One process per GPU (torch.distributed, backend nccl = RCCL);
        launched by the driver's torchrun line, or by this script itself: with --gpus N > 1 and no WORLD_SIZE in the
        environment the parent process (which never touches a GPU) starts N ranks and relays rank 0's line.
        Every rank scans ITS shard for the SAME 10 000-query batch; partial (key64, id) lists are all-gathered and
        merged on every rank.  Weak scaling (the corpus grows with N: N x 12.5M rows, 100M at N = 8); `value` counts
        the (query x shard) scans all ranks complete per second; `qps_whole_corpus` is queries/s against the N-shard
        corpus.

Output: ONE JSON line on rank 0 carrying `roofline` (dominant kernel, MFMA-bound, algorithmic flops 2*Q*N*D per
launch over the HIP-event time recorded on the search stream during the timed steps), and at N = 1 `cpu_baseline`
(threaded-BLAS GEMM expansion; `cpu_baselines` also holds the C port and the NumPy LinearSearcher
restatement), `qps_plugin_host_io` + `first_call_ms` (the reference harness's view: HipExactSearch.batch_search with
pageable NumPy in/out, experiment_runner.py:431-437) and `also.gaussian1m` (the non-fp16-exact case, same run).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "vectordb-retrieval_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

PEAK_F16_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_I8_TOPS = 5000.0      # dense int8 MFMA peak (2x fp16 per clock), same guide
HBM_PEAK_GBS = 8000.0
WORKLOADS = {
    #  name        (rows, dim, queries, k, metric, generator)
    "sift1m": (1_000_000, 128, 10_000, 10, "l2", "sift_like"),
    "gaussian1m": (1_000_000, 128, 10_000, 10, "l2", "gaussian"),
    "glove1.2m": (1_200_000, 50, 10_000, 10, "ip", "glove_like"),
    "marco2m": (2_000_000, 768, 10_000, 10, "ip", "gaussian"),   # MS MARCO-shaped shard slice (config 5 is 12.5M/GPU)
    # BASELINE configs[4] per-GPU shard (100M x 768 over 8 GPUs): rows generated ON DEVICE in fixed 500k-row
    # blocks seeded by the global block number, so the data do not depend on the number of ranks
    "marco12.5m": (12_500_000, 768, 10_000, 10, "ip", "device_gaussian"),
    "gauss50m": (50_000_000, 128, 10_000, 10, "l2", "device_gaussian"),     # capacity check of the flat D <= 128 path
    "smoke": (10_000, 128, 100, 10, "l2", "random_reference"),
    # BASELINE configs[3]: IVF-Flat over the sift1m data, nlist = 1024, --nprobe 8 / 32 / 128
    "ivf1024": (1_000_000, 128, 10_000, 10, "l2", "sift_like"),
}
DEVICE_BLOCK_ROWS = 500_000


def device_rows(n: int, d: int, rank: int, dev):
    """(n, d) float32 standard-normal rows generated on `dev`, block b of the whole corpus from seed 1234 + b."""
    import torch

    X = torch.empty((n, d), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    blocks_per_rank = -(-n // DEVICE_BLOCK_ROWS)
    for b in range(blocks_per_rank):
        lo, hi = b * DEVICE_BLOCK_ROWS, min(n, (b + 1) * DEVICE_BLOCK_ROWS)
        gen.manual_seed(1234 + rank * blocks_per_rank + b)
        X[lo:hi].normal_(generator=gen)
    return X


def device_check(X_t, q_t, I_t, k: int, metric: str, id_base: int, sample: int = 32, dist=None, world: int = 1) -> float:
    """Recall of the first `sample` queries against a float64 torch scan of the device-resident corpus (used for
    workloads too large to hand to the CPU oracle).  With world > 1 every rank scans ITS shard, the per-shard exact
    top-k (scores + global ids) are all-gathered and merged, and the merged list is what the result is compared with
    (the result holds neighbours from every shard)."""
    import torch

    q = q_t[:sample].double()
    best_v, best_i = None, None
    for lo in range(0, X_t.shape[0], 1_000_000):
        x = X_t[lo:lo + 1_000_000].double()
        s = q @ x.T if metric == "ip" else -((q * q).sum(1, keepdim=True) - 2.0 * (q @ x.T) + (x * x).sum(1)[None, :])
        v, i = torch.topk(s, k, dim=1)
        i = i + lo + id_base
        if best_v is None:
            best_v, best_i = v, i
        else:
            cv, ci = torch.cat([best_v, v], 1), torch.cat([best_i, i], 1)
            best_v, sel = torch.topk(cv, k, dim=1)
            best_i = torch.gather(ci, 1, sel)
    if dist is not None and world > 1:
        vs = [torch.empty_like(best_v) for _ in range(world)]
        ids = [torch.empty_like(best_i) for _ in range(world)]
        dist.all_gather(vs, best_v.contiguous())
        dist.all_gather(ids, best_i.contiguous())
        cv, ci = torch.cat(vs, 1), torch.cat(ids, 1)
        best_v, sel = torch.topk(cv, k, dim=1)
        best_i = torch.gather(ci, 1, sel)
    got = I_t[:sample].cpu().numpy()
    ref = best_i.cpu().numpy()
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(ref, got)]))


def real_sift():
    """(X, Q, GT, where) from real SIFT1M TEXMEX files under $VDBHIP_DATA, or None (there are none offline)."""
    from vdbhip import io

    root = os.environ.get("VDBHIP_DATA")
    if not root:
        return None
    for d in (Path(root), Path(root) / "sift", Path(root) / "sift1m"):
        base, query, gt = d / "sift_base.fvecs", d / "sift_query.fvecs", d / "sift_groundtruth.ivecs"
        if base.exists() and query.exists():
            X, Q = io.read_fvecs(base), io.read_fvecs(query)
            G = io.read_ivecs(gt) if gt.exists() else None
            return np.ascontiguousarray(X), np.ascontiguousarray(Q), G, str(d)
    return None


def make_data(name: str, rank: int):
    from vdbhip import datasets

    n, d, nq, k, metric, gen = WORKLOADS[name]
    if gen == "device_gaussian":
        Q = np.random.default_rng(1235).standard_normal((nq, d), dtype=np.float32)
        return None, Q, k, metric
    if gen == "sift_like":
        X = datasets._sift_rows(np.random.default_rng(1234 + 7919 * rank), n, d)
        Q = datasets._sift_rows(np.random.default_rng(1235), nq, d)
    elif gen == "gaussian":
        X = np.random.default_rng(1234 + 7919 * rank).standard_normal((n, d), dtype=np.float32)
        Q = np.random.default_rng(1235).standard_normal((nq, d), dtype=np.float32)
    elif gen == "glove_like":
        X = 0.5 * np.random.default_rng(50 + 7919 * rank).standard_normal((n, d), dtype=np.float32)
        Q = 0.5 * np.random.default_rng(51).standard_normal((nq, d), dtype=np.float32)
    else:
        X, Q = datasets.random_reference(d, n, nq, 42)
    return np.ascontiguousarray(X, np.float32), np.ascontiguousarray(Q, np.float32), k, metric


def recall_vs(ids_ref, ids_got, k) -> float:
    return float(np.mean([len(set(a[:k].tolist()) & set(b[:k].tolist())) / k for a, b in zip(ids_ref, ids_got)]))


def cpu_baselines(X, Q, k, metric, gpu_ids):
    """SURVEY 8(d) CPU legs on this box's host cores, each on a bounded query sample of the same workload; the ids of
    the BLAS leg double as a recall check of the GPU result."""
    from oracle import blas_baseline, c_oracle

    out = {}
    blas, ids = blas_baseline.time_gemm_expansion(X, Q, k, metric, budget_s=10.0)
    out["blas_gemm_expansion"] = blas
    recall = recall_vs(ids, gpu_ids[:len(ids)], k)
    # the hand-written C port (scalar / omp simd dot loops, no BLAS), all host threads
    c_oracle.build()
    cores = c_oracle.num_threads()
    probe = min(len(Q), 4 * cores)
    t0 = time.perf_counter()
    c_oracle.knn(X, Q[:probe], k, metric, mode=c_oracle.MODE_GEMM32)
    dt = time.perf_counter() - t0
    sample = int(min(len(Q), max(probe, probe * 6.0 / max(dt, 1e-6))))
    t0 = time.perf_counter()
    c_oracle.knn(X, Q[:sample], k, metric, mode=c_oracle.MODE_GEMM32)
    dt = time.perf_counter() - t0
    out["c_port"] = {"value": round(sample / dt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                     "impl": "oracle/knn_oracle.c MODE_GEMM32 (float32 expansion, OpenMP, no BLAS)",
                     "sample": f"first {sample} of {len(Q)} queries against all {len(X)} rows, {dt:.1f} s"}
    if metric == "l2":   # the reference's own CPU path for YAML `exact` (NumPy LinearSearcher), tiny query batches
        out["numpy_linear_searcher"] = blas_baseline.time_linear_searcher(X, Q, k, metric, qbatch=4, budget_s=5.0)
    return out, recall


def timed_device_loop(index, q_t, nq, k, D_t, I_t, stream, steps, warmup, torch):
    for _ in range(warmup):
        index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
    torch.cuda.synchronize()
    index.set_option("timing", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = index.stats()
    index.set_option("timing", 0)
    return elapsed, st


def scaling_reference_leg(vdbhip, torch, dev, local_rank, stream, steps, warmup):
    """The N > 1 default workload (one config-5 shard, 12.5M x 768 inner product) on ONE GPU through the sharded code
    path (partial lists -> packed buffer -> merge; the all-gather of a one-rank world is a copy): the N = 1 point of
    the weak-scaling series, in the same run as the headline."""
    name = "marco12.5m"
    n, d, nq, k, metric, _ = WORKLOADS[name]
    _, Q, _, _ = make_data(name, 0)
    X_t = device_rows(n, d, 0, dev)
    index = vdbhip.FlatIndex(d, metric, local_rank)
    index.add_device(X_t.data_ptr(), n, id_base=0)
    q_t = torch.from_numpy(Q).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    my_pack = torch.empty((2, nq, k), dtype=torch.int64, device=dev)
    all_pack = torch.empty((1, 2, nq, k), dtype=torch.int64, device=dev)

    def step():
        index.search_partial_device(q_t.data_ptr(), nq, k, my_pack[0].data_ptr(), my_pack[1].data_ptr(), stream)
        all_pack.copy_(my_pack.unsqueeze(0))
        vdbhip.merge_packed_partials_device(metric, local_rank, all_pack.data_ptr(), 1, nq, k,
                                            D_t.data_ptr(), I_t.data_ptr(), stream)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    index.set_option("timing", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = index.stats()
    leg = {"value": round(nq * steps / el, 1), "unit": "queries/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(el / steps * 1e3, 4),
           "config": f"{name}: {n} rows x {d} dims, {nq} queries, k={k}, {metric}; rows generated on device",
           "roofline": roofline_of(st, nq, n, d, name),
           "recall@10_vs_float64_torch_sample": round(device_check(X_t, q_t, I_t, k, metric, 0), 6),
           "note": "N = 1 point of the weak-scaling series: `bench.py --gpus N` (N > 1) runs this shard on every GPU "
                   "and reports value = N x queries / time, so value(N) / (N x this value) is the scaling efficiency"}
    index.close()
    return leg


def serving_leg(index, q_t, k, D_t, I_t, stream, torch, n, d):
    """Serving-shaped batches on the headline index: wall time of one search_device + synchronise (median of 30) for
    1 and 64 queries, and the HBM roofline of the scan -- a single query streams the whole scan copy once."""
    leg = {}
    for nq in (1, 64):
        for _ in range(5):
            index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
        torch.cuda.synchronize()
        index.set_option("timing", 1)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        st = index.stats()
        index.set_option("timing", 0)
        i8 = int(st.get("scan_dtype", 0)) == 1
        dpad = -(-d // 32) * 32 if i8 else -(-d // 16) * 16
        panel_bytes = float(n) * dpad * (1 if i8 else 2)
        scan_ms = float(st["last_scan_ms"])
        traffic, source = None, None
        try:
            ent = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get("sift1m_serving_nq1", {})
            if i8 and n == 1_000_000 and d == 128:
                traffic, source = ent.get("hbm_bytes_per_launch"), ent.get("source")
        except Exception:  # noqa: BLE001
            pass
        leg[f"nq{nq}"] = {"latency_us": round(float(np.median(ts)) * 1e6, 1), "scan_us": round(scan_ms * 1e3, 1),
                          "roofline": {"bound": "hbm", "achieved": round(panel_bytes / (scan_ms * 1e-3) / 1e9, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(panel_bytes / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "algorithmic_bytes_per_launch": panel_bytes, "traffic": traffic,
                                       "traffic_source": (source + " -- a recorded PMC pass, not measured in this run") if source else None,
                                       "note": "rows x padded dims x %d B: the %s scan copy read once" %
                                               (1 if i8 else 2, "int8" if i8 else "fp16")}}
    return leg


def roofline_of(st, nq, n, d, workload, ivf_rows_probed=None):
    """MFMA roofline of the dominant kernel from the HIP-event time the library recorded on the search stream."""
    scan_ms = float(st["last_scan_ms"])
    i8 = int(st.get("scan_dtype", 0)) == 1
    rows = float(n) if ivf_rows_probed is None else float(ivf_rows_probed)
    flops = 2.0 * nq * rows * d if ivf_rows_probed is None else 2.0 * rows * d
    peak = PEAK_I8_TOPS if i8 else PEAK_F16_TFLOPS
    achieved = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    if ivf_rows_probed is not None:
        kernel = "scan_kernel<ITEMS> (list-major IVF scan)"
    elif d <= 128:
        kernel = "scan_i8_kernel" if i8 else "scan_kernel<%d>" % (4 if d <= 64 else 8)
    else:
        kernel = "scan16_kloop_kernel"
    traffic, source = None, None
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    if pmc.exists():
        try:
            ent = json.loads(pmc.read_text()).get(workload + ("_i8" if i8 else ""), {})
            traffic, source = ent.get("hbm_bytes_per_launch"), ent.get("source")
        except Exception:  # noqa: BLE001
            pass
    return {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": peak,
            "unit": "TOP/s (int8)" if i8 else "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
            "traffic_source": (source + " -- a recorded PMC pass of an earlier run of this command, not measured in "
                               "this run") if source else None,
            "kernel_ms": round(scan_ms, 4), "pipeline_ms": round(float(st["last_total_ms"]), 4),
            "algorithmic_flops_per_launch": flops}


def launch_ranks(args) -> int:
    """--gpus N > 1 without a torchrun environment: this parent starts one child per GPU BEFORE it makes any GPU call
    (counting devices does not initialise one), relays rank 0's JSON line and fails if any rank fails."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {have}")
    port = int(os.environ.get("MASTER_PORT", 29400 + os.getpid() % 2000))
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        return 1
    lines = [ln for ln in out0.decode().splitlines() if ln.startswith("{")]
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 1
    print(lines[-1], flush=True)
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--nprobe", type=int, default=32, help="ivf1024 workload: lists probed per query")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip plugin host-I/O timing and the gaussian1m leg")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    workload = args.workload or ("sift1m" if args.gpus == 1 else "marco12.5m")

    # stdout carries exactly ONE line, the JSON result: everything native libraries print on file descriptor 1 while
    # the job runs (the pool exports NCCL_DEBUG=VERSION, so RCCL prints a five-line banner there) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch

    import vdbhip
    from vdbhip import _ffi

    if not torch.cuda.is_available() or _ffi.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    n, d = WORKLOADS[workload][:2]
    data_tag, GT = "synthetic", None
    real = real_sift() if workload in ("sift1m", "ivf1024") and world == 1 else None
    if real is not None:
        X, Q, GT, where = real
        k, metric = 10, "l2"
        n, d = X.shape
        data_tag = f"sift1m (real TEXMEX files from {where})"
    else:
        X, Q, k, metric = make_data(workload, rank)
    nq = Q.shape[0]
    X_t = device_rows(n, d, rank, dev) if X is None else None
    torch.cuda.synchronize()
    ivf = workload == "ivf1024"
    extras = {}
    t0 = time.perf_counter()
    if ivf:
        index = vdbhip.IVFFlatIndex(d, 1024, metric, local_rank)
        index.train(X, niter=25, seed=1234, max_points_per_centroid=256)
        extras["train_s"] = round(time.perf_counter() - t0, 3)
        index.add(X)
        index.set_nprobe(args.nprobe)
    elif X is None:
        index = vdbhip.FlatIndex(d, metric, local_rank)
        index.add_device(X_t.data_ptr(), n, id_base=rank * n)
        torch.cuda.synchronize()
    elif world == 1 and not args.no_extras:
        # the reference harness's view first, in a fresh process state (experiment_runner.py:330, 431-437: build_index,
        # then batch_search timed with time.time(), NO warm-up): pageable NumPy queries in, NumPy (D, I) out
        algo = vdbhip.get_algorithm_instance("HipExactSearch", d, name="bench", metric=metric, device=local_rank)
        algo.build_index(X)
        extras["build_s_plugin"] = round(time.perf_counter() - t0, 3)
        t1 = time.time()
        algo.batch_search(Q, k=k)
        extras["first_call_ms"] = round((time.time() - t1) * 1e3, 3)
        for _ in range(3):
            algo.batch_search(Q, k=k)
        times = []
        for _ in range(11):
            t1 = time.time()
            algo.batch_search(Q, k=k)
            times.append(time.time() - t1)
        med = float(np.median(times))
        extras["qps_plugin_host_io"] = round(nq / med, 1)
        extras["plugin_host_io_ms"] = round(med * 1e3, 4)
        extras["plugin_host_io_note"] = ("HipExactSearch.batch_search(Q (10000,128) pageable numpy) -> numpy (D, I): H2D of Q "
                                         "and D2H of the result inside the timed call; median of 11 after 3 warm-ups; "
                                         "first_call_ms = the very first call after build_index, no warm-up search "
                                         "(build_index sizes the workspace for 10 000 queries: vdb_reserve)")
        index = algo.index
    else:
        index = vdbhip.FlatIndex(d, metric, local_rank)
        index.add(X, id_base=rank * n)
    build_s = time.perf_counter() - t0

    stream = torch.cuda.current_stream().cuda_stream
    q_t = torch.from_numpy(Q).to(dev)
    D_t = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I_t = torch.empty((nq, k), dtype=torch.int64, device=dev)
    sharded = world > 1 or os.environ.get("VDBHIP_BENCH_FORCE_SHARDED") == "1"   # (rehearsal of the N>1 code path)
    if sharded:
        # one packed buffer per rank: keys (nq*k float64) immediately followed by ids (nq*k int64) -> ONE all-gather
        my_pack = torch.empty((2, nq, k), dtype=torch.int64, device=dev)
        all_pack = torch.empty((world, 2, nq, k), dtype=torch.int64, device=dev)

    def local_step():
        index.search_partial_device(q_t.data_ptr(), nq, k, my_pack[0].data_ptr(), my_pack[1].data_ptr(), stream)

    def step():
        if not sharded:
            index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
        else:
            local_step()
            if world > 1:
                dist.all_gather_into_tensor(all_pack, my_pack)
            else:
                all_pack.copy_(my_pack.unsqueeze(0))
            vdbhip.merge_packed_partials_device(metric, local_rank, all_pack.data_ptr(), world, nq, k,
                                                D_t.data_ptr(), I_t.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    shard_alone_ms = None
    if sharded:      # this rank's shard scan without the exchange (untimed region): what the collective + merge add
        t0 = time.perf_counter()
        for _ in range(3):
            local_step()
        torch.cuda.synchronize()
        shard_alone_ms = (time.perf_counter() - t0) / 3 * 1e3
        fence()
    index.set_option("timing", 1)      # HIP events around the scan kernel, on the search stream, per step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    st = index.stats()
    index.set_option("timing", 0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = world * nq * args.steps / elapsed
    rows_probed = None
    if ivf:
        rows_probed = float(st.get("last_rows_scanned", 0)) or nq * args.nprobe / 1024.0 * n
    roof = roofline_of(st, nq, n, d, workload, ivf_rows_probed=rows_probed)
    if ivf:      # the list scan is HBM-bound per query unless queries are grouped per list -- report both
        bytes_q = 4.0 * d * rows_probed            # float32 rows each (query, probe) pair would read un-grouped
        roof["hbm_equiv"] = {"bound": "hbm", "achieved": round(bytes_q / (roof["kernel_ms"] * 1e-3) / 1e9, 1),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "note": "4*D bytes x (query, row) pairs scanned / scan time: what a per-query list scan "
                                     "would have to stream; the list-major scan reads each list once per query group"}
    i8 = int(st.get("scan_dtype", 0)) == 1
    dtype = ("i8 MFMA scan (i32 accumulate) + f64 exact refine" if i8
             else "f16 MFMA scan (f32 accumulate) + f64 exact refine")

    if workload == "sift1m":
        metric_name = "QPS @ recall@10 (SIFT1M%s, 10k-query batch, k=10)" % ("" if real else "-shaped")
    elif ivf:
        metric_name = f"QPS (SIFT1M-shaped IVF-Flat nlist=1024 nprobe={args.nprobe}, k={k})"
    else:
        metric_name = f"QPS ({workload}, k={k})"
    out = {
        "metric": metric_name,
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": data_tag,
        "config": {"workload": f"{workload}: {n} rows x {d} dims per GPU, {nq} queries, k={k}, {metric}; "
                               + ("IVF-Flat nlist=1024 nprobe=%d, own k-means; " % args.nprobe if ivf
                                  else "brute-force exact k-NN; ") + "inputs resident in HBM",
                   "rows_per_gpu": n, "dim": d, "queries": nq, "k": k, "metric": metric,
                   "sharding": "none" if world == 1 else
                   f"row-sharded x{world} (weak scaling: corpus = {world} x {n} rows), RCCL all-gather of partial top-k"},
        "rccl_ranks": int(dist.get_world_size()) if world > 1 else 1,
        "roofline": roof,
        "pipeline": {"path": st["last_path_name"], "candidates_per_query": round(st["last_candidates"] / nq, 2),
                     "rescan_bins": int(st["last_rescan_bins"]), "fallback_queries": int(st["last_fallback_queries"]),
                     "corpus_fp16_exact": int(st["corpus_fp16_exact"]), "build_s": round(build_s, 3),
                     "hbm_resident_mb": round(st["bytes_resident"] / 2 ** 20, 1)},
    }
    out.update(extras)
    if world > 1 or sharded:
        out["qps_whole_corpus"] = round(nq * args.steps / elapsed, 1)
        out["shard_scan_alone_ms"] = round(shard_alone_ms, 4) if shard_alone_ms else None
        out["scaling_reference"] = ("the N = 1 point of this workload is `also[\"%s\"].value` of the `--gpus 1` line (or "
                                    "`--gpus 1 --workload %s`): the default N = 1 workload is sift1m, a different "
                                    "problem" % (workload, workload))

    gpu_ids = I_t.cpu().numpy()
    if GT is not None:
        out["recall@10_vs_sift_groundtruth"] = round(recall_vs(GT[:, :k], gpu_ids, k), 6)
    if ivf:
        flat = vdbhip.FlatIndex(d, metric, local_rank)
        flat.add(X)
        _, ie = flat.search(Q, k)
        flat.close()
        out["recall@10_vs_exact"] = round(recall_vs(ie, gpu_ids, k), 6)
        if not args.no_cpu_baseline:
            # CPU leg of config 4 ("vs FAISS-CPU": faiss is not installed, so the C restatement of the same IVF-Flat
            # search -- same centroids, same lists, OpenMP over queries -- on a bounded query sample; its ids double as
            # a bit-level check of the GPU result)
            from oracle import c_oracle

            c_oracle.build()
            C, lor = index.centroids(), index.assignment()
            probe = 256
            t1 = time.perf_counter()
            c_oracle.ivf_search(X, C, lor, Q[:probe], k, args.nprobe, metric)
            dt = time.perf_counter() - t1
            sample = int(min(nq, max(probe, probe * 8.0 / max(dt, 1e-6))))
            t1 = time.perf_counter()
            _, io_ = c_oracle.ivf_search(X, C, lor, Q[:sample], k, args.nprobe, metric)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(sample / dt, 2), "unit": "queries/s", "cores": c_oracle.num_threads(),
                                   "kind": "port", "impl": "oracle/ivf_oracle.c (canonical float64 list scan, OpenMP)",
                                   "sample": f"first {sample} of {nq} queries, nprobe {args.nprobe}, {dt:.1f} s"}
            out["ids_equal_cpu_oracle_sample"] = bool(np.array_equal(io_, gpu_ids[:sample]))
    if X is None:
        out["recall@10_vs_float64_torch_sample"] = round(device_check(X_t, q_t, I_t, k, metric, rank * n, dist=dist,
                                                                       world=world), 6)
    elif rank == 0 and world == 1 and not ivf:
        if workload == "sift1m" and not args.no_extras:
            # second timed workload of the same run: Gaussian 1M (corpus NOT exact in fp16 -> non-trivial guard)
            Xg, Qg, kg, mg = make_data("gaussian1m", 0)
            gi = vdbhip.FlatIndex(Xg.shape[1], mg, local_rank)
            gi.add(Xg)
            qg_t = torch.from_numpy(Qg).to(dev)
            el, sg = timed_device_loop(gi, qg_t, len(Qg), kg, D_t, I_t, stream, args.steps, args.warmup, torch)
            from oracle import c_oracle

            _, io_ = c_oracle.knn(Xg, Qg[:32], kg, mg)
            out["also"] = {"gaussian1m": {
                "value": round(len(Qg) * args.steps / el, 1), "unit": "queries/s",
                "ms_per_step": round(el / args.steps * 1e3, 4),
                "roofline": roofline_of(sg, len(Qg), Xg.shape[0], Xg.shape[1], "gaussian1m"),
                "candidates_per_query": round(sg["last_candidates"] / len(Qg), 2),
                "rescan_bins": int(sg["last_rescan_bins"]), "fallback_queries": int(sg["last_fallback_queries"]),
                "corpus_fp16_exact": int(sg["corpus_fp16_exact"]),
                "ids_equal_cpu_oracle_first32": bool(np.array_equal(I_t[:32].cpu().numpy(), io_))}}
            gi.close()
            del Xg, Qg
            out["also"]["serving"] = serving_leg(index, q_t, k, D_t, I_t, stream, torch, n, d)
            index.close()
            out["also"]["marco12.5m"] = scaling_reference_leg(vdbhip, torch, dev, local_rank, stream,
                                                              min(args.steps, 10), min(args.warmup, 2))
        if not args.no_cpu_baseline:
            legs, recall = cpu_baselines(X, Q, k, metric, gpu_ids)
            out["cpu_baseline"] = legs["blas_gemm_expansion"]
            out["cpu_baselines"] = legs
            out["recall@10_vs_cpu_blas_sample"] = round(recall, 6)
    if world > 1:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
