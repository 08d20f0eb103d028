"""N > 1 path on CPU: two gloo ranks drive vdbhip.sharded (shard arithmetic, all-gather layout, merge
contract).  The per-shard engine is a TEST DOUBLE built on the oracle -- the product engine is HIP only."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_shard_bounds_cover_rows_exactly():
    from vdbhip.sharded import shard_bounds

    for n in (0, 1, 7, 8, 9, 1000, 1_000_003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert a <= b == c <= d
            per = -(-n // world)
            assert all(b - a <= per for a, b in spans)


class _OracleEngine:
    """CPU stand-in for HipShardEngine (same interface) -- test infrastructure only."""

    def __init__(self, dim, metric, device):
        import torch

        from oracle import c_oracle

        self.torch, self.o, self.metric = torch, c_oracle, metric

    def add(self, x, id_base):
        self.x, self.id_base = np.ascontiguousarray(x), id_base

    def to_device(self, q):
        return self.torch.from_numpy(np.ascontiguousarray(q))

    def search_partial(self, q, k):
        qn = q.numpy()
        if len(self.x) == 0:
            keys = np.full((len(qn), k), np.inf)
            ids = np.full((len(qn), k), -1, np.int64)
        else:
            _, ids, keys = self.o.knn(self.x, qn, k, self.metric, id_base=self.id_base, return_keys=True)
        return self.torch.from_numpy(np.stack([keys.view(np.int64), ids]))      # packed (2, nq, k)

    def merge(self, all_pack):
        a = all_pack.numpy()
        return self.o.merge_partials(np.ascontiguousarray(a[:, 0]).view(np.float64), np.ascontiguousarray(a[:, 1]),
                                     self.metric)


class _OracleIVFEngine(_OracleEngine):
    """CPU stand-in for HipIVFShardEngine: k-means is replaced by a seeded row sample (the plumbing under test is
    train-on-rank-0 + broadcast + per-shard add + partial merge, not the clustering)."""

    def __init__(self, dim, metric, device, nlist):
        super().__init__(dim, metric, device)
        self.nlist, self.nprobe, self.trained_here = nlist, 1, False

    def train(self, x, **kw):
        self.trained_here = True
        self.C = np.ascontiguousarray(x[np.random.default_rng(kw.get("seed", 1234)).choice(len(x), self.nlist, replace=False)])
        return self.C

    def set_centroids(self, c):
        self.C = np.ascontiguousarray(c)

    def set_nprobe(self, nprobe):
        self.nprobe = int(nprobe)

    def search_partial(self, q, k):
        qn = q.numpy()
        lor = self.o.ivf_assign(self.C, self.x, self.metric)
        _, ids = self.o.ivf_search(self.x, self.C, lor, qn, k, self.nprobe, self.metric, id_base=self.id_base)
        keys = np.full(ids.shape, np.inf)
        for i in range(len(qn)):      # exact float64 keys of the rows found (partial contract: key +inf / id -1 padding)
            m = ids[i] >= 0
            if m.any():
                keys[i, m] = self.o.pair_keys(self.x, qn[i:i + 1], (ids[i, m] - self.id_base)[None, :], self.metric)[0]
        return self.torch.from_numpy(np.stack([keys.view(np.int64), ids]))


def _ivf_worker(rank, world, port, out_dir):
    sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from vdbhip.sharded import HipShardedApproximateSearch

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        X = rng.standard_normal((1500, 16)).astype(np.float32)
        Q = rng.standard_normal((9, 16)).astype(np.float32)
        for metric in ("l2", "ip"):
            algo = HipShardedApproximateSearch("ivf_sh", 16, "IVF12,Flat", metric=metric, nprobe=3, seed=5,
                                               engine_factory=_OracleIVFEngine)
            algo.build_index(X)
            d, i = algo.batch_search(Q, k=6)
            np.savez(Path(out_dir) / f"ivf_r{rank}_{metric}.npz", d=d, i=i, C=algo.centroids,
                     trained=np.array(algo.engine.trained_here))
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_sharded_ivf_equals_unsharded(tmp_path, oracle):
    import torch.multiprocessing as mp

    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_ivf_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(11)
    X = rng.standard_normal((1500, 16)).astype(np.float32)
    Q = rng.standard_normal((9, 16)).astype(np.float32)
    for metric in ("l2", "ip"):
        g0, g1 = (np.load(tmp_path / f"ivf_r{r}_{metric}.npz") for r in range(world))
        assert bool(g0["trained"]) and not bool(g1["trained"])       # trained once, on rank 0 ...
        np.testing.assert_array_equal(g0["C"], g1["C"])              # ... and broadcast
        C = g0["C"]
        d_ref, i_ref = oracle.ivf_search(X, C, oracle.ivf_assign(C, X, metric), Q, 6, 3, metric)
        for g in (g0, g1):
            np.testing.assert_array_equal(g["i"], i_ref)
            np.testing.assert_array_equal(g["d"], d_ref)


def _worker(rank, world, port, out_dir):
    sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from vdbhip.sharded import HipShardedExactSearch

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(3)
        X = rng.standard_normal((1001, 24)).astype(np.float32)
        Q = rng.standard_normal((13, 24)).astype(np.float32)
        for metric in ("l2", "ip"):
            algo = HipShardedExactSearch("sh", 24, metric=metric, engine_factory=_OracleEngine)
            algo.build_index(X)
            d, i = algo.batch_search(Q, k=7)
            d1, i1 = algo.search(Q[0], k=7)
            # per-rank loader: rows [lo, hi) of a memory-mapped .npy / of an .fvecs file, nothing else
            from vdbhip import io

            npy, fv = Path(out_dir) / f"corpus_{metric}.npy", Path(out_dir) / f"corpus_{metric}.fvecs"
            if rank == 0:
                np.save(npy, X)
                io.write_fvecs(fv, X)
            dist.barrier()
            res = {}
            for tag, path in (("npy", npy), ("fvecs", fv)):
                a2 = HipShardedExactSearch("sh_file", 24, metric=metric, engine_factory=_OracleEngine)
                a2.build_index_from_file(str(path))
                res[f"d_{tag}"], res[f"i_{tag}"] = a2.batch_search(Q, k=7)
                res[f"shard_{tag}"] = np.array(a2.shard)
                res[f"rows_{tag}"] = np.array(len(a2.engine.x))
            np.savez(Path(out_dir) / f"r{rank}_{metric}.npz", d=d, i=i, d1=d1, i1=i1,
                     shard=np.array(algo.shard), **res)
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_equal_unsharded(tmp_path, oracle):
    import torch.multiprocessing as mp

    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(3)
    X = rng.standard_normal((1001, 24)).astype(np.float32)
    Q = rng.standard_normal((13, 24)).astype(np.float32)
    for metric in ("l2", "ip"):
        d_ref, i_ref = oracle.knn(X, Q, 7, metric)
        shards = []
        for r in range(world):
            g = np.load(tmp_path / f"r{r}_{metric}.npz")
            np.testing.assert_array_equal(g["i"], i_ref)      # every rank holds the full, identical result
            np.testing.assert_array_equal(g["d"], d_ref)
            np.testing.assert_array_equal(g["i1"], i_ref[0])
            shards.append(tuple(g["shard"]))
            for tag in ("npy", "fvecs"):                      # the file loader gives the same shards and results
                np.testing.assert_array_equal(g[f"i_{tag}"], i_ref)
                np.testing.assert_array_equal(g[f"d_{tag}"], d_ref)
                assert tuple(g[f"shard_{tag}"]) == tuple(g["shard"])
                assert int(g[f"rows_{tag}"]) == g["shard"][1] - g["shard"][0]      # only its own rows were handed over
        assert shards == [(0, 501), (501, 1001)]


# ---- no silent world = 1 ----------------------------------------------------------------
def test_world_size_without_a_process_group_raises(monkeypatch):
    """WORLD_SIZE > 1 announced, no process group, no rendezvous variables: an error, never a silent full scan per rank."""
    from vdbhip.sharded import HipShardedApproximateSearch, HipShardedExactSearch, ensure_process_group

    for var in ("RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    X = np.zeros((16, 4), np.float32)
    with pytest.raises(RuntimeError, match="WORLD_SIZE=2 but no torch.distributed process group"):
        HipShardedExactSearch("sh", 4, engine_factory=_OracleEngine).build_index(X)
    with pytest.raises(RuntimeError, match="WORLD_SIZE=2"):
        HipShardedApproximateSearch("sh", 4, "IVF2,Flat", engine_factory=_OracleIVFEngine).build_index(X)
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert ensure_process_group() == (0, 1)
    monkeypatch.delenv("WORLD_SIZE")
    assert ensure_process_group() == (0, 1)


def _autoinit_worker(rank, world, port, out_dir):
    """What a rank of `torchrun ... scripts/run_full_benchmark.py` looks like to the plugin: the launcher's environment,
    nobody has called init_process_group."""
    sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), VDBHIP_DIST_BACKEND="gloo")
    import torch.distributed as dist

    from vdbhip.sharded import HipShardedExactSearch

    assert not dist.is_initialized()
    rng = np.random.default_rng(8)
    X = rng.standard_normal((777, 12)).astype(np.float32)
    Q = rng.standard_normal((5, 12)).astype(np.float32)
    algo = HipShardedExactSearch("sh", 12, metric="l2", engine_factory=_OracleEngine)
    algo.build_index(X)
    try:
        assert dist.is_initialized() and dist.get_backend() == "gloo" and (algo.rank, algo.world) == (rank, world)
        d, i = algo.batch_search(Q, k=4)
        np.savez(Path(out_dir) / f"auto_r{rank}.npz", d=d, i=i, shard=np.array(algo.shard))
    finally:
        dist.destroy_process_group()


def test_process_group_is_initialised_from_the_launcher_environment(tmp_path, oracle):
    import torch.multiprocessing as mp

    world, port = 2, 33500 + (os.getpid() % 2000)
    mp.spawn(_autoinit_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(8)
    X = rng.standard_normal((777, 12)).astype(np.float32)
    Q = rng.standard_normal((5, 12)).astype(np.float32)
    d_ref, i_ref = oracle.knn(X, Q, 4, "l2")
    for r in range(world):
        g = np.load(tmp_path / f"auto_r{r}.npz")
        np.testing.assert_array_equal(g["i"], i_ref)
        np.testing.assert_array_equal(g["d"], d_ref)
    assert tuple(np.load(tmp_path / "auto_r1.npz")["shard"]) == (389, 777)
