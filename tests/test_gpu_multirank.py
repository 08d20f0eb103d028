"""N > 1 with the PRODUCT engine: two ranks, each with its own HIP index, on the one GPU of the test box.

RCCL refuses two ranks on one device, so here the collective runs on gloo (staged through the host); the per-rank
HIP engine, the shard arithmetic, the packed partial layout, the device merge and the centroid broadcast are the
code the nccl run executes.  `tests/multirank_worker.py --backend nccl` under torchrun is the same check on a
multi-GPU node (not launchable from the one-GPU box).
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_two_ranks_share_the_gpu_hip_engine_equals_unsharded(tmp_path):
    sys.path.insert(0, str(ROOT / "tests"))
    import multirank_worker

    world, port = 2, 33500 + (os.getpid() % 2000)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(
            [sys.executable, str(ROOT / "tests" / "multirank_worker.py"), "--backend", "gloo", "--share-gpu", "--out",
             str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o[-4000:]}"
    multirank_worker.check(tmp_path, world)
