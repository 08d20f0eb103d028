"""bench.py's own N > 1 line, rehearsed with two ranks on the one GPU of the test box: the launcher, the
rendezvous, the packed all-gather (gloo, staged through the host: RCCL refuses two ranks on one device), the max-over-ranks
timing, the merged recall check and the strong-scaling checksum invariance all execute -- only the RCCL transport itself is
left to a multi-GPU node."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _bench(*args, env=None, timeout=600):
    e = dict(os.environ, **(env or {}))
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=e, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_strong_scaling_equal_one_rank():
    shared = {"VDBHIP_BENCH_SHARED_GPU": "1"}
    one = _bench("--gpus", "1", "--workload", "marco1m", "--scaling", "strong", "--steps", "3", "--warmup", "1",
                 env={"VDBHIP_BENCH_FORCE_SHARDED": "1"})
    two = _bench("--gpus", "2", "--workload", "marco1m", "--scaling", "strong", "--steps", "3", "--warmup", "1", env=shared)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["ranks"] == 2 and two["collective_backend"] == "gloo"
    assert two["scaling"] == "strong" and two["config"]["rows_per_gpu"] == 500_000 and two["config"]["corpus_rows"] == 1_000_000
    assert one["result_checksum"] == two["result_checksum"]          # same rows, same queries, same merged ids
    for line in (one, two):
        assert line["recall@10_vs_float64_torch_sample"] == 1.0
        assert line["exchange_ms"] > 0 and line["shard_scan_alone_ms"] > 0 and line["value"] > 0
        assert line["roofline"]["frac"] > 0 and set(line["roofline"]["stages_ms"]) == {"prep", "scan", "tail"}


def test_two_ranks_default_line_is_the_headline_workload_weak():
    """`--gpus 2` with no workload: the sift1m headline under weak scaling (every rank its own 1M x 128 shard), i.e. the same
    workload as the `--gpus 1` line, with the config-5 shape riding along (here skipped: --no-extras keeps the test short)."""
    two = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras", env={"VDBHIP_BENCH_SHARED_GPU": "1"})
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and "sift1m" in two["config"]["workload"]
    assert two["config"]["rows_per_gpu"] == 1_000_000 and two["config"]["corpus_rows"] == 2_000_000
    assert two["recall@10_vs_float64_torch_sample"] == 1.0
    assert two["value"] == pytest.approx(2 * two["qps_whole_corpus"], rel=1e-3)
    assert "i8" in two["dtype"]
