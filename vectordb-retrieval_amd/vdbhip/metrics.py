"""recall@k exactly as the reference evaluates it (src/benchmark/metrics.py:4-34): per query, the
overlap between the first k ground-truth ids and the first k returned ids over the size of the
ground-truth set, averaged over queries."""
from __future__ import annotations

import numpy as np


def recall_at_k(ground_truth: np.ndarray, predicted: np.ndarray, k: int) -> float:
    k = min(k, predicted.shape[1])
    hits = 0.0
    for gt_row, pred_row in zip(ground_truth, predicted):
        truth = set(gt_row[:k].tolist()) if ground_truth.shape[1] >= k else set(gt_row.tolist())
        if truth:
            hits += len(truth.intersection(pred_row[:k].tolist())) / len(truth)
    return hits / max(len(ground_truth), 1)


def latency_stats(timing_data) -> dict:
    """Latency summary with the keys of the reference's (unwired) `compute_cost_latency`
    (src/benchmark/metrics.py:212-237): mean / median / p95 / p99 / min / max of per-query seconds."""
    t = np.asarray(list(timing_data), dtype=np.float64)
    if t.size == 0:
        return {k: 0.0 for k in ("mean", "median", "p95", "p99", "min", "max")}
    return {"mean": float(t.mean()), "median": float(np.median(t)), "p95": float(np.percentile(t, 95)),
            "p99": float(np.percentile(t, 99)), "min": float(t.min()), "max": float(t.max())}
