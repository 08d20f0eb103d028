#!/usr/bin/env python3
"""bench.py -- QPS of the exact k-NN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sift1m|gaussian1m|glove1.2m|marco2m|marco12.5m|gauss50m|smoke]

A "step" is one pass of the hot path over one 10 000-query batch already resident in HBM:
libvdbhip's device pipeline (query prep -> fp16 MFMA scan + bin select -> exact float64 refine), and for
N > 1 the RCCL all-gather of the per-shard partial top-k plus the merge kernel.

N = 1   workload = BASELINE.json configs[1]: SIFT1M-shaped corpus (1 000 000 x 128, integer-valued float32),
        10 000 queries, k = 10, L2.  Synthetic (no dataset files exist offline); recipe in vdbhip/datasets.py.
N > 1   one process per GPU (torch.distributed, backend nccl = RCCL).  Every rank owns a SIFT1M-shaped
        1M-row shard of an N x 1M-row corpus (global ids = rank * 1M + row) and scans it for the SAME
        10 000-query batch; partial (key64, id) lists are all-gathered and merged on every rank.  Weak
        scaling: `value` counts the (query x 1M-row-shard) scans all ranks complete per second, which at N = 1
        is plain QPS on SIFT1M.

Output: ONE JSON line on rank 0 (contract in the task description) carrying `roofline` (dominant kernel =
scan_kernel, MFMA-bound, algorithmic flops 2*Q*N*D per launch over the HIP-event time recorded on the
search stream during the timed steps) and `cpu_baseline` (oracle/knn_oracle.c MODE_GEMM32, OpenMP on the
host cores, bounded query sample of the same workload; rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "vectordb-retrieval_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

PEAK_F16_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
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


def device_check(X_t, q_t, I_t, k: int, metric: str, id_base: int, sample: int = 32) -> float:
    """Recall of the first `sample` queries against a float64 torch scan of the device-resident corpus (used for
    workloads too large to hand to the CPU oracle)."""
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
    got = I_t[:sample].cpu().numpy()
    ref = best_i.cpu().numpy()
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(ref, got)]))


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


def cpu_baseline(X, Q, k, metric, gpu_ids, budget_s=15.0):
    """Time the CPU port (oracle/knn_oracle.c, MODE_GEMM32, all host cores) on a bounded query sample and
    use its ids to check the GPU result of the same queries."""
    from oracle import c_oracle

    c_oracle.build()
    cores = c_oracle.num_threads()
    probe = min(len(Q), 8 * cores)                       # enough query blocks to occupy every thread
    t0 = time.perf_counter()
    c_oracle.knn(X, Q[:probe], k, metric, mode=c_oracle.MODE_GEMM32)
    dt = time.perf_counter() - t0
    sample = int(min(len(Q), max(probe, probe * budget_s / max(dt, 1e-6))))
    sample = max(8 * cores, sample // (8 * cores) * (8 * cores)) if sample >= 8 * cores else sample
    sample = min(sample, len(Q))
    t0 = time.perf_counter()
    _, ids = c_oracle.knn(X, Q[:sample], k, metric, mode=c_oracle.MODE_GEMM32)
    dt = time.perf_counter() - t0
    hits = 0
    for a, b in zip(ids, gpu_ids[:sample]):
        hits += len(set(a.tolist()) & set(b.tolist()))
    return {
        "value": round(sample / dt, 2), "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": f"first {sample} of {len(Q)} queries against all {len(X)} rows, "
                  f"oracle/knn_oracle.c MODE_GEMM32 (float32 expansion, OpenMP), {dt:.1f} s",
    }, hits / float(sample * k)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sift1m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: everything native libraries print on file descriptor 1 while
    # the job runs (the pool exports NCCL_DEBUG=VERSION, so RCCL prints a five-line banner there) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch

    import vdbhip
    from vdbhip import _ffi

    if not torch.cuda.is_available() or _ffi.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    X, Q, k, metric = make_data(args.workload, rank)
    n, d = WORKLOADS[args.workload][:2]
    nq = Q.shape[0]
    X_t = device_rows(n, d, rank, dev) if X is None else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = vdbhip.FlatIndex(d, metric, local_rank)
    if X is None:
        index.add_device(X_t.data_ptr(), n, id_base=rank * n)
        torch.cuda.synchronize()
    else:
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

    def step():
        if not sharded:
            index.search_device(q_t.data_ptr(), nq, k, D_t.data_ptr(), I_t.data_ptr(), stream)
        else:
            index.search_partial_device(q_t.data_ptr(), nq, k, my_pack[0].data_ptr(), my_pack[1].data_ptr(), stream)
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
    scan_ms = float(st["last_scan_ms"])
    flops = 2.0 * nq * n * d
    achieved = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    traffic = None
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    if pmc.exists():
        try:
            traffic = json.loads(pmc.read_text()).get(args.workload, {}).get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None

    out = {
        "metric": "QPS @ recall@10 (SIFT1M-shaped, 10k-query batch, k=10)" if args.workload == "sift1m"
        else f"QPS ({args.workload}, k={k})",
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 MFMA scan (f32 accumulate) + f64 exact refine", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n} rows x {d} dims per GPU, {nq} queries, k={k}, {metric}; "
                               f"brute-force exact k-NN, inputs resident in HBM",
                   "rows_per_gpu": n, "dim": d, "queries": nq, "k": k, "metric": metric,
                   "sharding": "none" if world == 1 else f"row-sharded x{world}, RCCL all-gather of partial top-k"},
        "roofline": {"bound": "mfma", "kernel": ("scan_kernel<%d>" % (4 if d <= 64 else 8)) if d <= 128 else "scan16_kloop_kernel",
                     "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                     "kernel_ms": round(scan_ms, 4), "pipeline_ms": round(float(st["last_total_ms"]), 4),
                     "algorithmic_flops_per_launch": flops},
        "pipeline": {"path": st["last_path_name"], "candidates_per_query": round(st["last_candidates"] / nq, 2),
                     "rescan_bins": int(st["last_rescan_bins"]), "fallback_queries": int(st["last_fallback_queries"]),
                     "corpus_fp16_exact": int(st["corpus_fp16_exact"]), "build_s": round(build_s, 3),
                     "hbm_resident_mb": round(st["bytes_resident"] / 2 ** 20, 1)},
    }

    if X is None:
        out["recall@10_vs_float64_torch_sample"] = round(device_check(X_t, q_t, I_t, k, metric, rank * n), 6)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        gpu_ids = I_t.cpu().numpy()
        base, recall = cpu_baseline(X, Q, k, metric, gpu_ids)
        out["cpu_baseline"] = base
        out["recall@10_vs_cpu_oracle_sample"] = round(recall, 6)
    if world > 1:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
