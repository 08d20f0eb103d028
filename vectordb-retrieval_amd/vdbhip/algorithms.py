"""MI355X-backed plugins with the reference's names, argument meaning and output conventions.

  HipExactSearch         drop-in for ExactSearch            (src/algorithms/exact_search.py:6-78)
  HipBruteForceIndexer   drop-in for BruteForceIndexer      (src/algorithms/modular.py:121-133)
  HipLinearSearcher      drop-in for LinearSearcher         (src/algorithms/modular.py:312-390)

Convention matrix reproduced here; all return (float32 (Q,k), int64 (Q,k)), best first:
  ExactSearch     l2 -> squared L2          | cosine/ip -> raw inner product (NOT normalised), descending
                  k > N -> id -1, distance +FLT_MAX / -FLT_MAX   (faiss.IndexFlat padding)
  LinearSearcher  l2 -> sqrt(squared L2)    | cosine -> -(q^ . x^) | ip -> -(q . x), ascending
                  k > N -> id -1, distance +inf
The GPU work is done by libvdbhip through index.FlatIndex; nothing here computes distances on the CPU.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

from . import _ffi
from .index import FlatIndex
from .plugin_api import (BaseAlgorithm, BaseIndexer, BaseSearcher, IndexArtifact, Metadata, SearchResult,
                         register_algorithm, register_indexer, register_searcher)


def _safe_normalize(matrix: np.ndarray) -> np.ndarray:
    """Rows scaled to unit length, zero rows stay zero -- same NumPy calls as modular.py:109-111 so the
    normalised float32 operands are bit-identical to the reference's."""
    norms = np.linalg.norm(matrix, axis=1, keepdims=True)
    return np.divide(matrix, norms, out=np.zeros_like(matrix), where=norms > 0)


def _resolve_device(device, device_ids):
    """GPU(s) of an index: `device` (an ordinal, or a list of them) wins over `device_ids`; more than one ordinal means ONE
    index row-sharded over those GPUs inside this process (vdb_create_multi) -- the reference's harness is a single process
    (experiment_runner.py:329-331, 428-434), so this is how a YAML entry `{type: HipExactSearch, device_ids: [0, .., 7]}`
    reaches eight MI355X.  Returns an int or a list of ints."""
    from .index import normalize_devices

    if device is not None:
        return normalize_devices(device)
    if device_ids is not None and not isinstance(device_ids, (int, np.integer)) and len(device_ids) > 0:
        return normalize_devices(list(device_ids))
    if isinstance(device_ids, (int, np.integer)):
        return int(device_ids)
    import os

    return int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("VDBHIP_DEVICE_FROM_RANK") else 0


DEFAULT_RESERVE_QUERIES = 10_000


def reserve_workspace(index, params: dict, k: int = 10) -> None:
    """Size the search workspace at build time for batches of `reserve_queries` queries (default 10 000, 0 = leave it to
    the first search).  The reference harness times its first batch_search with the allocations it triggers
    (experiment_runner.py:431-437; no warm-up, metrics_methodology.md:119-121); FAISS' GPU resources reserve their scratch
    memory at construction for the same reason."""
    n = int((params or {}).get("reserve_queries", DEFAULT_RESERVE_QUERIES))
    if n > 0 and index is not None and index.ntotal > 0:
        index.reserve(n, k)


def apply_engine_options(index, params: dict) -> None:
    """`engine_options: {name: value}` of an algorithm / indexer / searcher config entry are handed to `vdb_set_option`
    before the corpus is added (include/vdbhip.h lists them; e.g. `stream_panels: 1` for a 768-dim corpus that should
    occupy ~1.2x instead of ~1.6x its float32 bytes).  An unknown name raises, as the C-ABI does."""
    for key, value in ((params or {}).get("engine_options") or {}).items():
        index.set_option(str(key), float(value))


class HipExactSearch(BaseAlgorithm):
    """Exact k-NN on one MI355X -- or, with `device_ids: [..]` of several, on ONE index row-sharded over them in this
    process; same constructor and results as ExactSearch (faiss.IndexFlat)."""

    def __init__(self, name: str, dimension: int, metric: str = "l2", device=None,
                 device_ids=None, **kwargs: Any) -> None:
        super().__init__(name, dimension, **kwargs)
        if device_ids is not None:
            self.config["device_ids"] = list(device_ids) if not isinstance(device_ids, (int, np.integer)) else int(device_ids)
        # exact_search.py:23 -- 'l2' -> METRIC_L2, anything else -> METRIC_INNER_PRODUCT (no normalisation)
        self.metric = "l2" if metric == "l2" else "ip"
        self.device = _resolve_device(device, device_ids)
        self.index: Optional[FlatIndex] = None

    def build_index(self, vectors: np.ndarray, metadata: Metadata = None) -> None:
        self.vectors = _ffi.as_f32_c(vectors)
        if self.vectors.ndim != 2 or self.vectors.shape[1] != self.dimension:
            raise ValueError(f"expected (n, {self.dimension}) vectors, got {self.vectors.shape}")
        self.index = FlatIndex(self.dimension, self.metric, self.device)
        apply_engine_options(self.index, self.config)
        self.index.add(self.vectors)
        reserve_workspace(self.index, self.config)
        self.index_built = True

    def _require_built(self) -> None:
        if not self.index_built:
            raise RuntimeError("Index has not been built yet.")

    def search(self, query: np.ndarray, k: int = 10) -> SearchResult:
        self._require_built()
        d, i = self.index.search(np.asarray(query, dtype=np.float32).reshape(1, -1), k)
        return d[0], i[0]

    def batch_search(self, queries: np.ndarray, k: int = 10) -> SearchResult:
        self._require_built()
        q = _ffi.as_f32_c(queries)
        self.record_operation("ndis", float(q.shape[0]) * float(self.index.ntotal))
        return self.index.search(q, k)

    def get_memory_usage(self) -> float:
        """MB resident in HBM (picked up by experiment_runner.py:493-497)."""
        return self.index.stats()["bytes_resident"] / (1024.0 * 1024.0) if self.index else 0.0


class HipBruteForceIndexer(BaseIndexer):
    """Keeps the raw matrix (as the reference does) and uploads it to HBM once, at build time."""

    def build(self, vectors: np.ndarray, metadata: Metadata = None) -> IndexArtifact:
        store = _ffi.as_f32_c(vectors)
        if store.ndim != 2 or store.shape[1] != self.dimension:
            raise ValueError("Vector dimension mismatch in HipBruteForceIndexer")
        device = _resolve_device(self.params.get("device"), self.params.get("device_ids"))
        if self.metric not in ("l2", "cosine", "ip"):
            # same late failure as the reference: an unknown metric only breaks at search time
            return IndexArtifact(kind="raw_vectors", data=store,
                                 metadata={"metric": self.metric, "normalize_vectors": False})
        upload = _safe_normalize(store) if self.metric == "cosine" else store
        index = FlatIndex(self.dimension, "l2" if self.metric == "l2" else "ip", device)
        apply_engine_options(index, self.params)
        index.add(upload)
        reserve_workspace(index, self.params)
        return IndexArtifact(kind="raw_vectors", data=store,
                             metadata={"metric": self.metric, "normalize_vectors": self.metric == "cosine",
                                       "hip_index": index, "hip_index_metric": self.metric})


class HipLinearSearcher(BaseSearcher):
    """LinearSearcher semantics; the scan itself runs on the MI355X."""

    def attach(self, artifact: IndexArtifact, vectors: np.ndarray, metadata: Metadata = None) -> None:
        if artifact.kind != "raw_vectors":
            raise ValueError("HipLinearSearcher requires 'raw_vectors' artifact")
        store = artifact.data
        if store.shape[1] != self.dimension:
            raise ValueError("Vector dimension mismatch in LinearSearcher")
        self._ntotal = int(store.shape[0])
        self._index: Optional[FlatIndex] = None
        meta = artifact.metadata or {}
        if self.metric in ("l2", "cosine", "ip"):
            if meta.get("hip_index") is not None and meta.get("hip_index_metric") == self.metric:
                self._index = meta["hip_index"]
            else:  # artifact from the reference's own BruteForceIndexer, or built for another metric
                device = _resolve_device(self.params.get("device"), self.params.get("device_ids"))
                data = _ffi.as_f32_c(store)
                self._index = FlatIndex(self.dimension, "l2" if self.metric == "l2" else "ip", device)
                apply_engine_options(self._index, self.params)
                self._index.add(_safe_normalize(data) if self.metric == "cosine" else data)
                reserve_workspace(self._index, self.params)
        self._prepared = True

    def _prepare_query(self, query: np.ndarray) -> np.ndarray:
        query = np.asarray(query)
        if query.ndim == 1:
            query = query.reshape(1, -1)
        return query.astype(np.float32, copy=True)

    def search(self, query: np.ndarray, k: int = 10) -> SearchResult:
        d, i = self.batch_search(self._prepare_query(query), k)
        return d[0], i[0]

    def batch_search(self, queries: np.ndarray, k: int = 10) -> SearchResult:
        if not self._prepared:
            raise RuntimeError("LinearSearcher not attached to an index")
        q = self._prepare_query(queries)
        if self.metric not in ("l2", "cosine", "ip"):
            raise ValueError(f"Unsupported metric '{self.metric}' for LinearSearcher")
        if self._ntotal == 0:
            raise RuntimeError("LinearSearcher cannot operate on empty index")
        if self.metric == "cosine":
            q = _safe_normalize(q)
        d, i = self._index.search(q, k)
        pad = i < 0
        if self.metric == "l2":
            d = np.sqrt(np.where(pad, np.float32(0), d), dtype=np.float32)
        else:
            d = -d
        if pad.any():
            d[pad] = np.inf
        return d.astype(np.float32, copy=False), i

    def get_memory_usage(self) -> float:
        return self._index.stats()["bytes_resident"] / (1024.0 * 1024.0) if getattr(self, "_index", None) else 0.0


def rerank_candidates(index: FlatIndex, queries: np.ndarray, candidate_ids: np.ndarray, k: int, metric: str):
    """Candidate re-scoring with the conventions of FaissSearcher._batch_search_lsh_rerank (modular.py:455-534):
    per query the valid (>= 0) candidate ids are re-scored exactly and the best k returned -- Euclidean distance
    for 'l2', negated score for 'cosine' / 'ip' (queries and the indexed vectors already normalised for cosine),
    unused slots padded with +inf / -1.  One batched launch replaces the reference's per-query Python loop."""
    d, i = index.rerank(queries, candidate_ids, k)
    pad = i < 0
    d = np.sqrt(np.where(pad, np.float32(0), d), dtype=np.float32) if metric == "l2" else -d
    d[pad] = np.inf
    return d.astype(np.float32, copy=False), i


register_algorithm("HipExactSearch", HipExactSearch)
register_indexer("HipBruteForceIndexer", HipBruteForceIndexer)
register_searcher("HipLinearSearcher", HipLinearSearcher)
