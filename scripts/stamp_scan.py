#!/usr/bin/env python3
"""Diagnostic build of the flat scan kernel (scan_variant 6): per-wave shader-cycle sums of the stage head,
MFMA phases, select phases and barrier waits.  Shares only: the stamped build is slower than the real one."""
import os; os.environ.setdefault('VDBHIP_LIBRARY', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vectordb-retrieval_amd', 'vdbhip', 'libvdbhip_ablations.so'))  # `make -C vectordb-retrieval_amd ablations`
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from vdbhip import _ffi
from bench import make_data
X, Q, k, metric = make_data(sys.argv[1] if len(sys.argv) > 1 else "sift1m", 0)
idx = vdbhip.FlatIndex(X.shape[1], metric, 0); idx.add(X)
idx.search(Q, k)
idx.set_option("scan_variant", 6)
for _ in range(3):
    idx.search(Q, k)
buf = np.zeros(2560 * 8 * 8 + 64, np.uint64); n = ctypes.c_int64(0)
_ffi.check(_ffi.load().vdb_debug_fetch_stamps(idx._h, buf.ctypes.data, buf.size, ctypes.byref(n)))
w = buf[: n.value].reshape(-1, 8).astype(np.float64)
w = w[w[:, 4] > 0]
for name, sel in (("early", w[:, 5] == 0), ("late", w[:, 5] == 1)):
    v = w[sel]; st = v[:, 6].mean(); tiles = st * 4
    print(f"{name}: waves={len(v)} stages/wave={st:.0f} total={v[:,4].mean():.0f} cyc | per stage: head={v[:,0].mean()/st:.0f} "
          f"mfma={v[:,1].mean()/st:.0f} ({v[:,1].mean()/tiles:.0f}/tile) select={v[:,2].mean()/st:.0f} ({v[:,2].mean()/tiles:.0f}/tile) "
          f"barrier={v[:,3].mean()/st:.0f} | sum/total={(v[:,:4].sum(1)/v[:,4]).mean():.3f}")
