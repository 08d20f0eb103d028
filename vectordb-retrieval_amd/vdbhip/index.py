"""Thin object wrappers over the libvdbhip C-ABI handles (flat index; IVF-Flat in ivf.py).

The returned conventions are those of faiss.IndexFlat, i.e. of the reference's ExactSearch
(exact_search.py:62-78): squared L2 ascending / raw inner product descending, int64 ids,
-1 / +-FLT_MAX padding when k exceeds the number of indexed rows.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _ffi

_METRICS = {"l2": _ffi.METRIC_L2, "ip": _ffi.METRIC_IP}


def normalize_devices(device):
    """int -> int; a sequence of GPU ordinals -> list (length one -> its only member)."""
    if isinstance(device, (list, tuple, np.ndarray)):
        devs = [int(d) for d in device]
        if not devs:
            raise ValueError("empty device list")
        return devs[0] if len(devs) == 1 else devs
    return int(device)


class FlatIndex:
    """Device-resident brute-force index (replaces faiss.IndexFlat(d, metric)).  `device`: one GPU ordinal, or a list of
    them -- ONE index row-sharded over those MI355X inside this process (vdb_create_multi): same calls, same results."""

    def __init__(self, dim: int, metric: str = "l2", device=0):
        if metric not in _METRICS:
            raise ValueError(f"metric must be 'l2' or 'ip', got {metric!r}")
        self.dim, self.metric, self.device = int(dim), metric, normalize_devices(device)
        self._lib = _ffi.load()
        self._h = _ffi.create_handle(self.dim, _METRICS[metric], self.device)
        self.ntotal = 0

    @property
    def devices(self):
        return list(self.device) if isinstance(self.device, list) else [self.device]

    def _sync_ntotal(self) -> None:
        """The row count is the library's: an add may have replaced rows instead of appending (e.g. after a failed add)."""
        s = _ffi.Stats()
        _ffi.check(self._lib.vdb_stats(self._handle(), ctypes.byref(s)))
        self.ntotal = int(s.ntotal)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vdb_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if not self._h:
            raise RuntimeError("index handle already destroyed")
        return self._h

    # -- build --------------------------------------------------------------------------------------
    def add(self, vectors: np.ndarray, id_base: int = 0) -> None:
        """Append rows (faiss.Index.add): row i of the index, counted over all adds, has id `id_base + i`; every add of
        one index passes the same `id_base`.  `reset()` empties the index."""
        x = _ffi.as_f32_c(vectors)
        if x.ndim != 2 or x.shape[1] != self.dim:
            raise ValueError(f"expected (n, {self.dim}) vectors, got {x.shape}")
        try:
            _ffi.check(self._lib.vdb_add(self._handle(), _ffi.ptr(x), x.shape[0], int(id_base)), build_time=True)
        finally:
            self._sync_ntotal()

    def add_device(self, dev_ptr: int, n: int, id_base: int = 0, stream: int = 0) -> None:
        try:
            _ffi.check(self._lib.vdb_add_device(self._handle(), dev_ptr, int(n), int(id_base), stream or None),
                       build_time=True)
        finally:
            self._sync_ntotal()

    def reset(self) -> None:
        """Drop every row (faiss.Index.reset)."""
        _ffi.check(self._lib.vdb_reset(self._handle()), build_time=True)
        self._sync_ntotal()

    # -- search -------------------------------------------------------------------------------------
    def search(self, queries: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _ffi.as_f32_c(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise RuntimeError(f"expected (nq, {self.dim}) queries, got {q.shape}")
        nq = q.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        _ffi.check(self._lib.vdb_search(self._handle(), _ffi.ptr(q), nq, int(k), _ffi.ptr(D), _ffi.ptr(I)))
        return D, I

    def search_device(self, q_ptr: int, nq: int, k: int, d_ptr: int, i_ptr: int, stream: int = 0) -> None:
        """All pointers are device memory on this index's GPU; asynchronous on `stream`."""
        _ffi.check(self._lib.vdb_search_device(self._handle(), q_ptr, int(nq), int(k), d_ptr, i_ptr, stream or None))

    def search_partial_device(self, q_ptr: int, nq: int, k: int, keys_ptr: int, ids_ptr: int, stream: int = 0) -> None:
        _ffi.check(self._lib.vdb_search_partial_device(self._handle(), q_ptr, int(nq), int(k), keys_ptr, ids_ptr,
                                                       stream or None))

    def rerank(self, queries: np.ndarray, candidate_ids: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """Exact top-k among explicit candidate ids per query (nq, ncand) int64, -1 = empty slot."""
        q = _ffi.as_f32_c(queries)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        c = np.ascontiguousarray(candidate_ids, dtype=np.int64)
        if c.ndim != 2 or c.shape[0] != q.shape[0] or q.shape[1] != self.dim:
            raise RuntimeError(f"expected queries (nq, {self.dim}) and candidates (nq, ncand), got {q.shape}, {c.shape}")
        D = np.empty((q.shape[0], k), np.float32)
        I = np.empty((q.shape[0], k), np.int64)
        _ffi.check(self._lib.vdb_rerank(self._handle(), _ffi.ptr(q), q.shape[0], _ffi.ptr(c), c.shape[1], int(k),
                                        _ffi.ptr(D), _ffi.ptr(I)))
        return D, I

    # -- introspection --------------------------------------------------------------------------------
    def stats(self) -> dict:
        s = _ffi.Stats()
        _ffi.check(self._lib.vdb_stats(self._handle(), ctypes.byref(s)))
        return s.as_dict()

    def reserve(self, nq: int, k: int = 10) -> None:
        """Size the search workspace for batches of up to `nq` queries now (vdb_reserve): the reference harness times its
        very first batch_search, allocations included (experiment_runner.py:431-437)."""
        _ffi.check(self._lib.vdb_reserve(self._handle(), int(nq), int(k)), build_time=True)

    def set_option(self, key: str, value: float) -> None:
        _ffi.check(self._lib.vdb_set_option(self._handle(), key.encode(), float(value)), build_time=True)

    def debug_scan_scores(self, queries: np.ndarray, row0: int, nrows: int):
        q = _ffi.as_f32_c(queries)
        nq = q.shape[0]
        scores = np.empty((nq, nrows), np.float32)
        eps = np.empty((nq,), np.float32)
        cs = ctypes.c_double(0.0)
        _ffi.check(self._lib.vdb_debug_scan_scores(self._handle(), _ffi.ptr(q), nq, int(row0), int(nrows),
                                                   _ffi.ptr(scores), _ffi.ptr(eps), ctypes.byref(cs)))
        return scores, eps, float(cs.value)


def merge_partials_device(metric: str, device: int, keys_ptr: int, ids_ptr: int, nparts: int, nq: int, k: int,
                          d_ptr: int, i_ptr: int, stream: int = 0) -> None:
    """Merge (nparts, nq, k) per-shard partial lists (device memory) into the final (nq, k) result."""
    _ffi.check(_ffi.load().vdb_merge_partials_device(_METRICS[metric], int(device), keys_ptr, ids_ptr, int(nparts),
                                                     int(nq), int(k), d_ptr, i_ptr, stream or None))


def merge_packed_partials_device(metric: str, device: int, packed_ptr: int, nparts: int, nq: int, k: int, d_ptr: int,
                                 i_ptr: int, stream: int = 0) -> None:
    """Merge per-part packed buffers (nparts, 2, nq, k) of 8-byte words: keys then ids (one all-gather)."""
    _ffi.check(_ffi.load().vdb_merge_packed_partials_device(_METRICS[metric], int(device), packed_ptr, int(nparts),
                                                            int(nq), int(k), d_ptr, i_ptr, stream or None))
