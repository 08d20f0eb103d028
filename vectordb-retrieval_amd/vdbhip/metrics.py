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
