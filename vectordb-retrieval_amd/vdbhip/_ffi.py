"""ctypes binding of libvdbhip.so (include/vdbhip.h) -- the only way the Python side reaches the GPU.

There is NO CPU fallback: if the shared library is missing or no MI355X is present the calls fail
loudly (ImportError at load, RuntimeError from the library).  Status codes are mapped so that
build-time argument errors surface as ValueError (as the reference does, modular.py:316-320) while
every search-time failure is a RuntimeError -- the reference's harness swallows ValueError/TypeError
from batch_search and silently degrades to per-query search (experiment_runner.py:442-446).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p
from pathlib import Path
from typing import Optional

import numpy as np

VDB_OK, VDB_ERR_INVALID, VDB_ERR_STATE, VDB_ERR_HIP, VDB_ERR_NOMEM, VDB_ERR_UNSUPPORTED = range(6)
METRIC_L2, METRIC_IP = 0, 1
PATH_NAMES = {0: "none", 1: "exact_scan", 2: "mfma_scan", 3: "ivf"}

_LIB_NAME = "libvdbhip.so"
_lib: Optional[ctypes.CDLL] = None


class Stats(Structure):
    _fields_ = [
        ("ntotal", c_int64), ("dim", c_int32), ("metric", c_int32), ("bytes_resident", c_int64),
        ("last_path", c_int32), ("corpus_fp16_exact", c_int32), ("last_nq", c_int64),
        ("last_candidates", c_int64), ("last_rescan_bins", c_int64), ("last_fallback_queries", c_int64),
        ("last_scan_ms", c_float), ("last_total_ms", c_float), ("nlist", c_int32), ("nprobe", c_int32),
        ("scan_dtype", c_int32), ("has_i8_copy", c_int32), ("last_rows_scanned", c_int64),
        ("upload_blocks", c_int64),
        ("graph_replays", c_int64),
        ("ndevices", c_int32), ("scan_shape", c_int32), ("bytes_workspace", c_int64), ("last_prep_ms", c_float), ("last_tail_ms", c_float),
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["last_path_name"] = PATH_NAMES.get(self.last_path, "?")
        return d


# every symbol include/vdbhip.h declares: (restype, argtypes)
SIGNATURES = {
    "vdb_abi_version": (c_int, []),
    "vdb_last_error": (c_char_p, []),
    "vdb_device_count": (c_int, [POINTER(c_int)]),
    "vdb_create": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "vdb_create_multi": (c_int, [c_int, c_int, POINTER(c_int), c_int, POINTER(c_void_p)]),
    "vdb_destroy": (c_int, [c_void_p]),
    "vdb_reset": (c_int, [c_void_p]),
    "vdb_add": (c_int, [c_void_p, c_void_p, c_int64, c_int64]),
    "vdb_add_device": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "vdb_search": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "vdb_search_device": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "vdb_search_partial_device": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "vdb_merge_partials_device": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p,
                                          c_void_p, c_void_p]),
    "vdb_merge_packed_partials_device": (c_int, [c_int, c_int, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p,
                                                 c_void_p]),
    "vdb_rerank": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vdb_rerank_device": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vdb_ivf_train": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_uint64, c_int]),
    "vdb_ivf_set_centroids": (c_int, [c_void_p, c_void_p, c_int]),
    "vdb_ivf_get_centroids": (c_int, [c_void_p, c_void_p]),
    "vdb_ivf_add": (c_int, [c_void_p, c_void_p, c_int64, c_int64]),
    "vdb_ivf_add_assigned": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "vdb_ivf_set_nprobe": (c_int, [c_void_p, c_int]),
    "vdb_ivf_get_assignment": (c_int, [c_void_p, c_void_p]),
    "vdb_ivf_search": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "vdb_ivf_search_device": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "vdb_ivf_search_partial_device": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "vdb_reserve": (c_int, [c_void_p, c_int64, c_int]),
    "vdb_stats": (c_int, [c_void_p, POINTER(Stats)]),
    "vdb_set_option": (c_int, [c_void_p, c_char_p, c_double]),
    "vdb_debug_fetch_stamps": (c_int, [c_void_p, c_void_p, c_int64, POINTER(c_int64)]),
    "vdb_debug_scan_scores": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                      POINTER(c_double)]),
}


def library_path() -> Path:
    override = os.environ.get("VDBHIP_LIBRARY")
    return Path(override) if override else Path(__file__).resolve().parent / _LIB_NAME


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
    libhsa-runtime64.so.1 (same sonames as /opt/rocm's), and whichever copy is loaded first serves every
    later user.  torch only works on top of ITS copy, so when torch is installed but not imported yet the
    bundled runtime is loaded first and libvdbhip.so binds to it (this is also what happens when the caller
    imports torch before vdbhip).  Set VDBHIP_SYSTEM_HIP=1 to bind to /opt/rocm instead (torch-free processes)."""
    import sys

    if "torch" in sys.modules or os.environ.get("VDBHIP_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if bundled.exists():
        try:
            ctypes.CDLL(str(bundled), mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> ctypes.CDLL:
    """dlopen libvdbhip.so and attach the prototypes (no GPU is touched by loading)."""
    global _lib
    if _lib is None:
        path = library_path()
        _share_hip_runtime_with_torch()
        if not path.exists():
            raise ImportError(
                f"{path} not found: build it with `make -C vectordb-retrieval_amd` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        lib = ctypes.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.vdb_abi_version() != 4:
            raise ImportError("libvdbhip.so ABI version mismatch")
        _lib = lib
    return _lib


def last_error() -> str:
    msg = load().vdb_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


class VdbError(RuntimeError):
    """Failure reported by libvdbhip (never ValueError/TypeError: see module docstring)."""


def check(status: int, *, build_time: bool = False) -> None:
    if status == VDB_OK:
        return
    msg = last_error() or f"libvdbhip status {status}"
    if build_time and status == VDB_ERR_INVALID:
        raise ValueError(msg)
    if status == VDB_ERR_NOMEM:
        raise MemoryError(msg)
    raise VdbError(msg)


def create_handle(dim: int, metric: int, device) -> c_void_p:
    """`device`: one GPU ordinal -> vdb_create; a sequence of ordinals -> vdb_create_multi (ONE index row-sharded over those
    GPUs inside this process; an ordinal may repeat).  A sequence of length one is the single-device index."""
    h = c_void_p()
    lib = load()
    if isinstance(device, (list, tuple, np.ndarray)):
        devs = [int(d) for d in device]
        if not devs:
            raise ValueError("empty device list")
        if len(devs) == 1:
            check(lib.vdb_create(int(dim), int(metric), devs[0], ctypes.byref(h)), build_time=True)
        else:
            arr = (c_int * len(devs))(*devs)
            check(lib.vdb_create_multi(int(dim), int(metric), arr, len(devs), ctypes.byref(h)), build_time=True)
    else:
        check(lib.vdb_create(int(dim), int(metric), int(device), ctypes.byref(h)), build_time=True)
    return h


def device_count() -> int:
    n = c_int(0)
    check(load().vdb_device_count(ctypes.byref(n)))
    return int(n.value)


def as_f32_c(a: np.ndarray) -> np.ndarray:
    """float32 + C-contiguous without needless copies (the idiom of exact_search.py:34-37, 76-77;
    accepts memmaps, float64, Fortran order, fancy-indexed copies)."""
    if isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]:
        return a
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a: Optional[np.ndarray]) -> Optional[int]:
    return None if a is None else a.ctypes.data
