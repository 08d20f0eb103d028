"""Row-sharded exact search over the GPUs of one node: one process per GPU, RCCL all-gather of partials.

The reference has no distributed code; this is the MI355X-native scaling path of its brute-force
hot path.  Rank r of W keeps rows [r*ceil(N/W), min(N, (r+1)*ceil(N/W))) in its own HBM, every
rank scans its shard for the SAME query batch, the per-shard partial top-k (float64 order keys + global ids,
`vdb_search_partial_device`, packed into one buffer) are exchanged with ONE all-gather, and every rank merges
them with `vdb_merge_packed_partials_device`.  The merge key (float64 key, id) makes the result independent of W.

`HipShardedApproximateSearch` shards IVF-Flat the same way: the coarse quantizer is trained on rank 0
and broadcast (the one other collective on the path), every rank files ITS rows under the shared centroids, probes
the same lists and contributes `vdb_ivf_search_partial_device` partials to the same all-gather + merge, so the
result equals the unsharded IVF index built on those centroids.

The collective and the shard arithmetic live here; the per-shard engine is pluggable so the N > 1 plumbing is
testable on CPU with the gloo backend (tests/test_sharded_gloo.py injects a CPU engine built on the oracle --
the product default is the HIP engine below and there is no CPU fallback).
"""
from __future__ import annotations

from typing import Any, Callable, Optional, Tuple

import numpy as np

from . import _ffi
from .plugin_api import BaseAlgorithm, Metadata, SearchResult, register_algorithm


def ensure_process_group(backend: Optional[str] = None, device: Optional[int] = None) -> Tuple[int, int]:
    """(rank, world) of the torch.distributed job; initialises the default group from the launcher's environment when the
    launcher started more than one rank and nobody has done it yet; raises RuntimeError when WORLD_SIZE > 1 is announced but
    there is neither a group nor the rendezvous variables to make one.  Never world = 1 by accident."""
    import os

    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    try:
        env_world = int(os.environ.get("WORLD_SIZE", "1") or "1")
    except ValueError:
        env_world = 1
    if env_world <= 1:
        return 0, 1
    missing = [v for v in ("RANK", "MASTER_ADDR", "MASTER_PORT") if not os.environ.get(v)]
    if missing or not dist.is_available():
        raise RuntimeError(
            f"WORLD_SIZE={env_world} but no torch.distributed process group is initialised and {', '.join(missing) or 'torch.distributed'} "
            "is missing: a row-sharded index would silently scan the whole corpus on every rank.  Launch with torchrun "
            "(or call torch.distributed.init_process_group first), or use device_ids=[...] of HipExactSearch for the "
            "single-process multi-GPU index.")
    backend = backend or os.environ.get("VDBHIP_DIST_BACKEND") or "nccl"
    kwargs = {}
    if backend == "nccl":
        import torch

        local = int(device if device is not None else os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend=backend, **kwargs)
    return dist.get_rank(), dist.get_world_size()


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: [rank*ceil(n/world), min(n, (rank+1)*ceil(n/world)))."""
    per = -(-n // world) if world > 0 else n
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


class HipShardEngine:
    """Per-rank engine: FlatIndex on this rank's GPU, torch tensors as device buffers."""

    def __init__(self, dim: int, metric: str, device: int):
        import torch

        from .index import FlatIndex

        self.torch = torch
        self.metric, self.device = metric, device
        torch.cuda.set_device(device)
        self.dev = torch.device("cuda", device)
        self.index = FlatIndex(dim, metric, device)

    def add(self, x: np.ndarray, id_base: int) -> None:
        self.index.add(x, id_base=id_base)

    def to_device(self, q: np.ndarray):
        return self.torch.from_numpy(q).to(self.dev)

    def search_partial(self, q_dev, k: int):
        """Packed partial of this shard: int64 tensor (2, nq, k) = [float64 keys (bit pattern) | ids]."""
        t = self.torch
        nq = q_dev.shape[0]
        pack = t.empty((2, nq, k), dtype=t.int64, device=self.dev)
        self.index.search_partial_device(q_dev.data_ptr(), nq, k, pack[0].data_ptr(), pack[1].data_ptr(),
                                         t.cuda.current_stream().cuda_stream)
        return pack

    def merge(self, all_pack) -> Tuple[np.ndarray, np.ndarray]:
        from .index import merge_packed_partials_device

        t = self.torch
        parts, _, nq, k = all_pack.shape
        D = t.empty((nq, k), dtype=t.float32, device=self.dev)
        I = t.empty((nq, k), dtype=t.int64, device=self.dev)
        merge_packed_partials_device(self.metric, self.device, all_pack.data_ptr(), parts, nq, k, D.data_ptr(),
                                     I.data_ptr(), t.cuda.current_stream().cuda_stream)
        return D.cpu().numpy(), I.cpu().numpy()


def all_gather_partials(pack, world: int):
    """(2, nq, k) per rank -> (world, 2, nq, k) on every rank: ONE collective (RCCL over xGMI for GPU tensors)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return pack.unsqueeze(0)
    out = torch.empty((world,) + tuple(pack.shape), dtype=pack.dtype, device=pack.device)
    if pack.is_cuda and dist.get_backend() == "nccl":       # the product path: RCCL, device buffers, one collective
        dist.all_gather_into_tensor(out, pack.contiguous())
    else:  # gloo (CPU tests, and the shared-GPU rehearsal of tests/test_gpu_multirank.py): staged through the host
        host = pack.contiguous().cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        out.copy_(torch.stack(parts))
    return out


class HipShardedExactSearch(BaseAlgorithm):
    """ExactSearch semantics (exact_search.py:6-78) over a corpus row-sharded across the ranks of the
    current torch.distributed job.  `build_index(vectors)` takes the FULL corpus array on every rank, as the
    reference's single-process harness would pass it -- a float32 `np.memmap` costs no host RAM, every rank only
    touches the pages of its own row block.  `build_index_from_file(path)` is the per-rank loader for corpora that
    are never materialised whole (100M x 768 = 307 GB): the file is memory-mapped and only rows [lo, hi) are read.
    Every rank passes the same queries and gets the full result back."""

    def __init__(self, name: str, dimension: int, metric: str = "l2", device: Optional[int] = None,
                 engine_factory: Optional[Callable[[int, str, int], Any]] = None, **kwargs: Any) -> None:
        super().__init__(name, dimension, **kwargs)
        self.metric = "l2" if metric == "l2" else "ip"      # exact_search.py:23
        self._device = device
        self._engine_factory = engine_factory
        self.engine = None
        self.rank, self.world = 0, 1
        self.ntotal = 0

    def _dist(self):
        """(rank, world) of the job this process belongs to.  A launcher that started several ranks (WORLD_SIZE > 1 in the
        environment) but no process group is NOT served as world = 1 -- every rank would scan the whole corpus on its own GPU
        and report nothing: with the launcher's rendezvous variables present (torchrun sets RANK / WORLD_SIZE / MASTER_ADDR /
        MASTER_PORT) the group is initialised here (backend: kwarg `dist_backend`, $VDBHIP_DIST_BACKEND, default nccl = RCCL,
        bound to this rank's GPU); without them it is an error."""
        return ensure_process_group(self.config.get("dist_backend"), self._device)

    def build_index(self, vectors: np.ndarray, metadata: Metadata = None) -> None:
        self.rank, self.world = self._dist()
        x = _ffi.as_f32_c(vectors)
        if x.ndim != 2 or x.shape[1] != self.dimension:
            raise ValueError(f"expected (n, {self.dimension}) vectors, got {x.shape}")
        self.ntotal = int(x.shape[0])
        lo, hi = shard_bounds(self.ntotal, self.world, self.rank)
        self._build_shard(x[lo:hi], lo, hi)

    def build_index_from_file(self, path: str, limit: Optional[int] = None, metadata: Metadata = None) -> None:
        """Per-rank loader: `path` is a 2-D float32 `.npy` (memory-mapped, dataset.py:376-471's cache format) or a
        TEXMEX `.fvecs` file; this rank reads rows [lo, hi) of it and nothing else."""
        from . import io

        self.rank, self.world = self._dist()
        if str(path).endswith(".fvecs"):
            raw = np.memmap(path, dtype=np.int32, mode="r")
            dim = int(raw[0]) if raw.size else self.dimension
            if dim != self.dimension or raw.size % (dim + 1) != 0:
                raise ValueError(f"{path}: not a .fvecs file of dimension {self.dimension}")
            rec = raw.reshape(-1, dim + 1)
            n = rec.shape[0] if limit is None else min(int(limit), rec.shape[0])
            lo, hi = shard_bounds(n, self.world, self.rank)
            x = np.ascontiguousarray(rec[lo:hi, 1:]).view(np.float32)
        else:
            arr = io.open_npy_rows(path, limit)
            if arr.shape[1] != self.dimension:
                raise ValueError(f"expected (n, {self.dimension}) vectors, got {arr.shape}")
            n = int(arr.shape[0])
            lo, hi = shard_bounds(n, self.world, self.rank)
            x = arr[lo:hi]                                   # still a memmap: pages are read during the upload
        self.ntotal = n
        self._build_shard(_ffi.as_f32_c(x), lo, hi)

    def _build_shard(self, rows: np.ndarray, lo: int, hi: int) -> None:
        import os

        device = self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", self.rank))
        factory = self._engine_factory or HipShardEngine
        self.engine = factory(self.dimension, self.metric, device)
        self.engine.add(rows, lo)
        self.shard = (lo, hi)
        self.index_built = True

    def batch_search(self, queries: np.ndarray, k: int = 10) -> SearchResult:
        if not self.index_built:
            raise RuntimeError("Index has not been built yet.")
        q = _ffi.as_f32_c(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        pack = self.engine.search_partial(self.engine.to_device(q), int(k))
        return self.engine.merge(all_gather_partials(pack, self.world))

    def search(self, query: np.ndarray, k: int = 10) -> SearchResult:
        d, i = self.batch_search(np.asarray(query, dtype=np.float32).reshape(1, -1), k)
        return d[0], i[0]


class HipIVFShardEngine(HipShardEngine):
    """Per-rank IVF engine: IVFFlatIndex on this rank's GPU filled with this rank's rows."""

    def __init__(self, dim: int, metric: str, device: int, nlist: int):
        import torch

        from .ivf import IVFFlatIndex

        self.torch = torch
        self.metric, self.device = metric, device
        torch.cuda.set_device(device)
        self.dev = torch.device("cuda", device)
        self.index = IVFFlatIndex(dim, nlist, metric, device)

    def train(self, x: np.ndarray, **kw) -> np.ndarray:
        self.index.train(x, **kw)
        return self.index.centroids()

    def set_centroids(self, c: np.ndarray) -> None:
        self.index.set_centroids(c)

    def set_nprobe(self, nprobe: int) -> None:
        self.index.set_nprobe(nprobe)


def broadcast_array(arr: Optional[np.ndarray], shape, world: int, src: int = 0) -> np.ndarray:
    """float32 array of `shape` from rank `src` to every rank (centroids: nlist x dim, a few hundred KiB)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return np.ascontiguousarray(arr, np.float32)
    use_cuda = dist.get_backend() == "nccl"
    t = torch.empty(shape, dtype=torch.float32) if arr is None else torch.from_numpy(np.ascontiguousarray(arr, np.float32))
    if use_cuda:
        t = t.cuda()
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


class HipShardedApproximateSearch(HipShardedExactSearch):
    """ApproximateSearch semantics ("IVF<nlist>,Flat", approximate_search.py:12-87) over a row-sharded corpus."""

    def __init__(self, name: str, dimension: int, index_type: str, metric: str = "l2", device: Optional[int] = None,
                 engine_factory: Optional[Callable[..., Any]] = None, **kwargs: Any) -> None:
        super().__init__(name, dimension, metric=metric, device=device, engine_factory=engine_factory, **kwargs)
        from .ivf import parse_ivf_key

        self.index_type = index_type
        self.nlist = parse_ivf_key(index_type)

    def build_index(self, vectors: np.ndarray, metadata: Metadata = None) -> None:
        import os

        self.rank, self.world = self._dist()
        x = _ffi.as_f32_c(vectors)
        if x.ndim != 2 or x.shape[1] != self.dimension:
            raise ValueError(f"expected (n, {self.dimension}) vectors, got {x.shape}")
        self.ntotal = int(x.shape[0])
        lo, hi = shard_bounds(self.ntotal, self.world, self.rank)
        device = self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", self.rank))
        factory = self._engine_factory or HipIVFShardEngine
        self.engine = factory(self.dimension, self.metric, device, self.nlist)
        cfg = self.config
        centroids = None
        if self.rank == 0:      # approximate_search.py:44: index.train(vectors) -- on the WHOLE corpus, once
            centroids = self.engine.train(x, niter=int(cfg.get("niter", 25)), seed=int(cfg.get("seed", 1234)),
                                          max_points_per_centroid=int(cfg.get("max_points_per_centroid", 256)))
        centroids = broadcast_array(centroids, (self.nlist, self.dimension), self.world)
        if self.rank != 0:
            self.engine.set_centroids(centroids)
        self.centroids = centroids
        self.engine.add(x[lo:hi], lo)
        if "nprobe" in cfg:     # approximate_search.py:50-51
            self.engine.set_nprobe(int(cfg["nprobe"]))
        self.shard = (lo, hi)
        self.index_built = True


register_algorithm("HipShardedExactSearch", HipShardedExactSearch)
register_algorithm("HipShardedApproximateSearch", HipShardedApproximateSearch)
