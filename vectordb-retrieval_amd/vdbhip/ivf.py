"""IVF-Flat on the MI355X behind the reference's three IVF entry points.

  HipApproximateSearch   drop-in for ApproximateSearch        (src/algorithms/approximate_search.py:6-87)
                         -- `index_type` keys of the form "IVF<nlist>,Flat" only (PQ/SQ codecs are out of scope,
                         SURVEY 2 row 6); 'l2' -> squared L2, anything else -> raw inner product; no sign flips.
  HipIVFIndexer          drop-in for FaissFactoryIndexer / FaissIVFIndexer (modular.py:224-309)
                         -- cosine = normalise + inner product (:253-262), runtime `nprobe` (:269-275)
  HipIVFSearcher         drop-in for FaissSearcher on IVF artifacts (modular.py:393-449, 536-548)
                         -- searcher `nprobe` overrides the indexer's (:437-441), queries normalised when the
                         artifact says so (:447-448), distances negated for cosine/ip (:545-546).

k-means is this library's own Lloyd iteration (FAISS's is not reproducible without FAISS), defaults mirroring
FAISS: 25 iterations, seed 1234, at most 256 training points per centroid.
"""
from __future__ import annotations

import ctypes
import re
from typing import Any, Optional, Tuple

import numpy as np

from . import _ffi
from .algorithms import _resolve_device, _safe_normalize, reserve_workspace
from .plugin_api import (BaseAlgorithm, BaseIndexer, BaseSearcher, IndexArtifact, Metadata, SearchResult,
                         register_algorithm, register_indexer, register_searcher)

_IVF_KEY = re.compile(r"^\s*IVF(\d+)\s*,\s*Flat\s*$")


def parse_ivf_key(key: str) -> int:
    m = _IVF_KEY.match(str(key))
    if not m:
        raise ValueError(f"unsupported index key {key!r}: only 'IVF<nlist>,Flat' is implemented on the HIP backend")
    return int(m.group(1))


class IVFFlatIndex:
    """Device-resident IVF-Flat index (replaces faiss.index_factory(d, "IVFn,Flat", metric))."""

    def __init__(self, dim: int, nlist: int, metric: str = "l2", device=0):
        """`device`: one GPU ordinal or a list of them (rows of every list split over those GPUs, coarse quantizer
        replicated: vdb_create_multi)."""
        from .index import normalize_devices

        if metric not in ("l2", "ip"):
            raise ValueError(f"metric must be 'l2' or 'ip', got {metric!r}")
        self.dim, self.nlist, self.metric, self.device = int(dim), int(nlist), metric, normalize_devices(device)
        self._lib = _ffi.load()
        self._h = _ffi.create_handle(self.dim, 0 if metric == "l2" else 1, self.device)
        self.is_trained = False
        self.ntotal = 0
        self.nprobe = 1

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vdb_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def train(self, x: np.ndarray, niter: int = 25, seed: int = 1234, max_points_per_centroid: int = 256) -> None:
        x = _ffi.as_f32_c(x)
        _ffi.check(self._lib.vdb_ivf_train(self._h, self.nlist, _ffi.ptr(x), x.shape[0], int(niter), int(seed),
                                           int(max_points_per_centroid)), build_time=True)
        self.is_trained = True
        self.ntotal = 0             # (new centroids: rows filed under the old ones are dropped)

    def set_centroids(self, centroids: np.ndarray) -> None:
        c = _ffi.as_f32_c(centroids)
        if c.shape != (self.nlist, self.dim):
            raise ValueError(f"expected ({self.nlist}, {self.dim}) centroids, got {c.shape}")
        _ffi.check(self._lib.vdb_ivf_set_centroids(self._h, _ffi.ptr(c), self.nlist), build_time=True)
        self.is_trained = True
        self.ntotal = 0

    def centroids(self) -> np.ndarray:
        out = np.empty((self.nlist, self.dim), np.float32)
        _ffi.check(self._lib.vdb_ivf_get_centroids(self._h, _ffi.ptr(out)))
        return out

    def add(self, x: np.ndarray, id_base: int = 0, list_of_row: Optional[np.ndarray] = None) -> None:
        """File the rows under the centroids, appending to the lists (faiss.IndexIVF.add; same `id_base` on every add of
        one index, `reset()` empties it).  `list_of_row` (int32, one list id per row, as `assignment()` returned it
        for this corpus and these centroids) skips the nearest-centroid pass: what loading a persisted index does."""
        x = _ffi.as_f32_c(x)
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected (n, {self.dim}) vectors, got {x.shape}")
        if list_of_row is None:
            _ffi.check(self._lib.vdb_ivf_add(self._h, _ffi.ptr(x), x.shape[0], int(id_base)), build_time=True)
        else:
            lor = np.ascontiguousarray(list_of_row, dtype=np.int32)
            if lor.shape != (x.shape[0],):
                raise ValueError(f"expected {x.shape[0]} list ids, got {lor.shape}")
            _ffi.check(self._lib.vdb_ivf_add_assigned(self._h, _ffi.ptr(x), x.shape[0], int(id_base), _ffi.ptr(lor)),
                       build_time=True)
        self.ntotal = int(self.stats()["ntotal"])      # (the library's count: an add may replace instead of append)

    def reset(self) -> None:
        """Drop every row; the centroids stay (faiss.IndexIVF.reset)."""
        _ffi.check(self._lib.vdb_reset(self._h), build_time=True)
        self.ntotal = 0

    def assignment(self) -> np.ndarray:
        out = np.empty((self.ntotal,), np.int32)
        _ffi.check(self._lib.vdb_ivf_get_assignment(self._h, _ffi.ptr(out)))
        return out

    def set_nprobe(self, nprobe: int) -> None:
        _ffi.check(self._lib.vdb_ivf_set_nprobe(self._h, int(nprobe)), build_time=True)
        self.nprobe = int(nprobe)

    def search(self, queries: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _ffi.as_f32_c(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise RuntimeError(f"expected (nq, {self.dim}) queries, got {q.shape}")
        D = np.empty((q.shape[0], k), np.float32)
        I = np.empty((q.shape[0], k), np.int64)
        _ffi.check(self._lib.vdb_ivf_search(self._h, _ffi.ptr(q), q.shape[0], int(k), _ffi.ptr(D), _ffi.ptr(I)))
        return D, I

    def search_device(self, q_ptr: int, nq: int, k: int, d_ptr: int, i_ptr: int, stream: int = 0) -> None:
        _ffi.check(self._lib.vdb_ivf_search_device(self._h, q_ptr, int(nq), int(k), d_ptr, i_ptr, stream or None))

    def search_partial_device(self, q_ptr: int, nq: int, k: int, keys_ptr: int, ids_ptr: int, stream: int = 0) -> None:
        """Per-shard partial top-k (float64 order keys + global ids) among the probed lists; device pointers."""
        _ffi.check(self._lib.vdb_ivf_search_partial_device(self._h, q_ptr, int(nq), int(k), keys_ptr, ids_ptr,
                                                           stream or None))

    def stats(self) -> dict:
        s = _ffi.Stats()
        _ffi.check(self._lib.vdb_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def reserve(self, nq: int, k: int = 10) -> None:
        """Size the search workspace for batches of up to `nq` queries now (vdb_reserve)."""
        _ffi.check(self._lib.vdb_reserve(self._h, int(nq), int(k)), build_time=True)

    def set_option(self, key: str, value: float) -> None:
        _ffi.check(self._lib.vdb_set_option(self._h, key.encode(), float(value)), build_time=True)


def _fingerprint(vectors: np.ndarray, centroids: np.ndarray, list_of_row: np.ndarray) -> dict:
    """What a persisted index records about its three files so that load_index can tell a vectors / centroids file that
    does not belong to list_of_row.npy (regenerated corpus, partial overwrite): SHA-256 of the centroids and of the lists
    in full, and of up to 4096 evenly spaced corpus rows (hashing a memory-mapped 38 GB shard in full is not cheap)."""
    import hashlib

    n = int(vectors.shape[0])
    pick = np.unique(np.linspace(0, max(n - 1, 0), num=min(n, 4096)).astype(np.int64)) if n else np.zeros(0, np.int64)
    rows = np.ascontiguousarray(np.asarray(vectors[pick], dtype=np.float32))
    return {"centroids": hashlib.sha256(np.ascontiguousarray(centroids, np.float32).tobytes()).hexdigest(),
            "list_of_row": hashlib.sha256(np.ascontiguousarray(list_of_row, np.int32).tobytes()).hexdigest(),
            "vectors_sample": hashlib.sha256(rows.tobytes()).hexdigest(), "vectors_sample_rows": int(len(pick)),
            "vectors_shape": [n, int(vectors.shape[1]) if vectors.ndim == 2 else 0]}


def _lists_match_sample(vectors, centroids, list_of_row, metric: str, rows: int = 512) -> bool:
    """Artifacts written before the fingerprint existed: re-assign a few hundred rows against the stored centroids
    (float64; a row within 1e-9 relative of a tie may sit in either list) and compare with the stored lists."""
    n = int(vectors.shape[0])
    if n == 0:
        return True
    pick = np.unique(np.linspace(0, n - 1, num=min(n, rows)).astype(np.int64))
    x = np.asarray(vectors[pick], dtype=np.float64)
    c = np.asarray(centroids, dtype=np.float64)
    score = -(x @ c.T) if metric == "ip" else (c * c).sum(1)[None, :] - 2.0 * (x @ c.T)
    best = score.min(axis=1)
    stored = score[np.arange(len(pick)), np.asarray(list_of_row)[pick]]
    return bool(np.all(stored - best <= 1e-9 * np.maximum(1.0, np.abs(best))))


def _build_ivf(vectors: np.ndarray, dim: int, key: str, metric: str, device: int, params: dict) -> IVFFlatIndex:
    index = IVFFlatIndex(dim, parse_ivf_key(key), metric, device)
    index.train(vectors, niter=int(params.get("niter", 25)), seed=int(params.get("seed", 1234)),
                max_points_per_centroid=int(params.get("max_points_per_centroid", 256)))
    index.add(vectors)
    return index


class HipApproximateSearch(BaseAlgorithm):
    """ApproximateSearch semantics for "IVF<nlist>,Flat": train -> add -> nprobe from kwargs; raw FAISS
    conventions (no normalisation, no sign flip)."""

    def __init__(self, name: str, dimension: int, index_type: str, metric: str = "l2", device: Optional[int] = None,
                 **kwargs: Any) -> None:
        super().__init__(name, dimension, **kwargs)
        self.index_type = index_type
        self.metric = "l2" if metric == "l2" else "ip"      # approximate_search.py:25
        self.device = _resolve_device(device, kwargs.get("device_ids"))
        self.index: Optional[IVFFlatIndex] = None
        parse_ivf_key(index_type)                            # fail at construction, like a bad factory string

    def build_index(self, vectors: np.ndarray, metadata: Metadata = None) -> None:
        self.vectors = np.asarray(vectors).astype(np.float32)
        self.index = _build_ivf(self.vectors, self.dimension, self.index_type, self.metric, self.device, self.config)
        self.index_built = True
        if "nprobe" in self.config:
            self.index.set_nprobe(int(self.config["nprobe"]))
        reserve_workspace(self.index, self.config)

    def search(self, query: np.ndarray, k: int = 10) -> SearchResult:
        if not self.index_built:
            raise RuntimeError("Index has not been built yet.")
        d, i = self.index.search(np.array([query], dtype=np.float32), k)
        return d[0], i[0]

    def batch_search(self, queries: np.ndarray, k: int = 10) -> SearchResult:
        if not self.index_built:
            raise RuntimeError("Index has not been built yet.")
        return self.index.search(np.asarray(queries).astype(np.float32), k)

    def get_memory_usage(self) -> float:
        return self.index.stats()["bytes_resident"] / (1024.0 * 1024.0) if self.index else 0.0

    # ---- persistence through the BaseAlgorithm hook (base_algorithm.py:98-120) ---------------------------
    # Layout and protocol follow the reference's only implementation (covertree_v2_2.py:101-182, 184-282):
    # temp dir + manifest.json + WRITE_COMPLETE sentinel written last + atomic rename; load refuses an
    # incomplete artifact or a manifest that does not match this instance.
    _FORMAT = "vdbhip-ivfflat-v1"

    def save_index(self, artifact_dir: str, context=None):
        import json
        import shutil
        import tempfile
        from pathlib import Path

        if not self.index_built or self.index is None:
            raise RuntimeError("Cannot persist HipApproximateSearch before build_index has completed.")
        context = context or {}
        target = Path(artifact_dir)
        target.parent.mkdir(parents=True, exist_ok=True)
        if target.exists():
            if not bool(context.get("force_rebuild", False)):
                raise FileExistsError(f"Artifact directory already exists: {target}. "
                                      "Set persistence.force_rebuild=true to overwrite.")
            shutil.rmtree(target)
        tmp = Path(tempfile.mkdtemp(prefix=f".{target.name}.tmp.", dir=str(target.parent)))
        try:
            centroids, lists = self.index.centroids(), self.index.assignment()
            np.save(tmp / "vectors.npy", self.vectors, allow_pickle=False)
            np.save(tmp / "centroids.npy", centroids, allow_pickle=False)
            np.save(tmp / "list_of_row.npy", lists, allow_pickle=False)
            build_metrics = dict(context.get("build_metrics", {}))
            manifest = {"format": self._FORMAT, "algorithm": type(self).__name__, "dimension": self.dimension,
                        "index_type": self.index_type, "metric": self.metric, "nlist": self.index.nlist,
                        "nprobe": self.index.nprobe, "n_vectors": int(self.index.ntotal),
                        "config_hash": context.get("config_hash"),
                        "sha256": _fingerprint(self.vectors, centroids, lists),
                        "files": {"vectors": "vectors.npy", "centroids": "centroids.npy",
                                  "list_of_row": "list_of_row.npy"}}
            (tmp / "manifest.json").write_text(json.dumps(manifest, indent=2), encoding="utf-8")
            (tmp / "build_metrics.json").write_text(json.dumps(build_metrics, indent=2), encoding="utf-8")
            (tmp / "WRITE_COMPLETE").write_text("ok\n", encoding="utf-8")
            tmp.rename(target)
        except Exception:
            shutil.rmtree(tmp, ignore_errors=True)
            raise
        return {"artifact_dir": str(target), "manifest_path": str(target / "manifest.json"),
                "build_time_s": float(build_metrics.get("build_time_s", 0.0) or 0.0)}

    def load_index(self, artifact_dir: str, context=None):
        import json
        from pathlib import Path

        path = Path(artifact_dir)
        if not path.is_dir():
            raise FileNotFoundError(f"Persisted HipApproximateSearch artifact directory not found: {path}")
        if not (path / "WRITE_COMPLETE").is_file():
            raise FileNotFoundError(f"Artifact is incomplete or corrupted (missing WRITE_COMPLETE): {path}")
        manifest = json.loads((path / "manifest.json").read_text(encoding="utf-8"))
        for key, want in (("format", self._FORMAT), ("dimension", self.dimension), ("index_type", self.index_type),
                          ("metric", self.metric)):
            if manifest.get(key) != want:
                raise ValueError(f"Persisted index mismatch for '{key}': artifact has {manifest.get(key)!r}, "
                                 f"this instance expects {want!r}")
        expected_hash = (context or {}).get("config_hash")
        if expected_hash and manifest.get("config_hash") and manifest["config_hash"] != expected_hash:
            raise ValueError("Persisted index was built with a different configuration (config_hash mismatch)")
        vectors = np.load(path / manifest["files"]["vectors"], mmap_mode="r")
        centroids = np.load(path / manifest["files"]["centroids"])
        self.vectors = vectors
        self.index = IVFFlatIndex(self.dimension, int(manifest["nlist"]), self.metric, self.device)
        self.index.set_centroids(centroids)       # no k-means: the stored quantizer is reused ...
        stored = np.load(path / manifest["files"]["list_of_row"])
        if stored.shape != (vectors.shape[0],) or (len(stored) and (stored.min() < 0 or stored.max() >= int(manifest["nlist"]))):
            raise ValueError("Persisted inverted lists do not match the persisted corpus")
        # the three files must be the ones that were written together: a corpus or quantizer that does not belong to the
        # stored lists would load silently and answer with degraded recall
        want = manifest.get("sha256")
        if want:
            have = _fingerprint(vectors, centroids, stored)
            bad = [key for key in ("vectors_shape", "vectors_sample", "centroids", "list_of_row") if have[key] != want.get(key)]
            if bad:
                raise ValueError("Persisted index files do not belong together (fingerprint mismatch: " + ", ".join(bad) + ")")
        elif not _lists_match_sample(vectors, centroids, stored, self.metric):
            raise ValueError("Persisted inverted lists do not match the persisted corpus and centroids")
        self.index.add(vectors, list_of_row=stored)   # ... and so are the stored lists: no assignment pass either
        self.index.set_nprobe(int(self.config.get("nprobe", manifest.get("nprobe", 1))))
        self.index_built = True
        metrics = {}
        bm = path / "build_metrics.json"
        if bm.is_file():
            metrics = json.loads(bm.read_text(encoding="utf-8"))
        return {"artifact_dir": str(path), "manifest_path": str(path / "manifest.json"),
                "build_time_s": float(metrics.get("build_time_s", 0.0) or 0.0)}


class HipIVFIndexer(BaseIndexer):
    """FaissFactoryIndexer / FaissIVFIndexer semantics for IVF-Flat keys."""

    _RESERVED = {"index_key", "index_type", "device", "device_ids", "niter", "seed", "max_points_per_centroid"}

    def __init__(self, name: str, dimension: int, metric: str = "l2", index_type: Optional[str] = None,
                 index_key: Optional[str] = None, **kwargs: Any) -> None:
        key = index_key or index_type or "IVF100,Flat"
        params = dict(kwargs)
        params.setdefault("index_type", key)
        super().__init__(name, dimension, metric, **params)
        self.index_key = self.index_type = key
        parse_ivf_key(key)

    def build(self, vectors: np.ndarray, metadata: Metadata = None) -> IndexArtifact:
        data = _ffi.as_f32_c(vectors)
        meta = {"metric": self.metric, "index_key": self.index_key, "faiss_metric": "l2"}
        metric = "l2"
        if self.metric == "cosine":
            data = _safe_normalize(data)
            metric = "ip"
            meta.update({"faiss_metric": "ip", "normalize_queries": True, "normalize_vectors": True})
        elif self.metric == "ip":
            metric = "ip"
            meta["faiss_metric"] = "ip"
        device = _resolve_device(self.params.get("device"), self.params.get("device_ids"))
        index = _build_ivf(data, self.dimension, self.index_key, metric, device, self.params)
        if "nprobe" in self.params:                       # runtime attribute of the index (modular.py:269-275)
            index.set_nprobe(int(self.params["nprobe"]))
            meta["nprobe"] = self.params["nprobe"]
        reserve_workspace(index, self.params)
        return IndexArtifact(kind="hip_ivf", data=index, metadata=meta)


class HipIVFSearcher(BaseSearcher):
    """FaissSearcher semantics over a HipIVFIndexer artifact."""

    def __init__(self, name: str, dimension: int, metric: str = "l2", **kwargs: Any) -> None:
        super().__init__(name, dimension, metric, **kwargs)
        self.index: Optional[IVFFlatIndex] = None
        self.normalize_queries = False

    def attach(self, artifact: IndexArtifact, vectors: np.ndarray, metadata: Metadata = None) -> None:
        if artifact.kind != "hip_ivf":
            raise ValueError("HipIVFSearcher requires 'hip_ivf' artifact")
        self.index = artifact.data
        meta = artifact.metadata or {}
        self.metric = meta.get("metric", self.metric)
        self.normalize_queries = meta.get("normalize_queries", False)
        self._prepared = True
        nprobe = self.params.get("nprobe")
        if nprobe is None:
            nprobe = meta.get("nprobe")
        if nprobe is not None:
            self.index.set_nprobe(int(nprobe))

    def _prepare_query(self, query: np.ndarray) -> np.ndarray:
        query = np.asarray(query)
        if query.ndim == 1:
            query = query.reshape(1, -1)
        query = query.astype(np.float32, copy=True)
        return _safe_normalize(query) if self.normalize_queries else query

    def search(self, query: np.ndarray, k: int = 10) -> SearchResult:
        d, i = self.batch_search(self._prepare_query(query), k)
        return d[0], i[0]

    def batch_search(self, queries: np.ndarray, k: int = 10) -> SearchResult:
        if not self._prepared:
            raise RuntimeError("FaissSearcher not attached to an index")
        d, i = self.index.search(self._prepare_query(queries), k)
        if self.metric in {"cosine", "ip"}:
            d = -d
        return d.astype(np.float32), i.astype(np.int64)

    def get_memory_usage(self) -> float:
        return self.index.stats()["bytes_resident"] / (1024.0 * 1024.0) if self.index else 0.0


register_algorithm("HipApproximateSearch", HipApproximateSearch)
register_algorithm("HipIVFFlat", HipApproximateSearch)
register_indexer("HipIVFIndexer", HipIVFIndexer)
register_indexer("HipFactoryIndexer", HipIVFIndexer)
register_searcher("HipIVFSearcher", HipIVFSearcher)
