#!/usr/bin/env python3
"""Diagnostic build of the int8 flat scan (i8_variant + 16, ablations library): where the cycles of scan_i8x16_kernel<2,8,8>
(default) or, with argument 32, of scan_i8_kernel<4,8,4> (option flat_shape = 32) go.  Per wave the kernel sums shader cycles (s_memtime) over the phases of every stage -- head (next stage's LDS-DMA issue
+ first fragment reads), MFMA phases (until the matrix pipe has delivered), select phases (+ the next tile's fragment
reads), tail (bin flush + bias store), barrier wait -- and reads s_memrealtime over the same span: the in-kernel clock is
cycles / ticks x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  Results stay exact; the stamped kernel is slower
than the production one (shares, not absolute times)."""
import os; os.environ.setdefault('VDBHIP_LIBRARY', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vectordb-retrieval_amd', 'vdbhip', 'libvdbhip_ablations.so'))  # `make -C vectordb-retrieval_amd ablations`
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "vectordb-retrieval_amd")]
import numpy as np, torch, vdbhip
from vdbhip import _ffi
from bench import make_data
X, Q, k, metric = make_data("sift1m", 0)
shape = int(sys.argv[1]) if len(sys.argv) > 1 else 16
idx = vdbhip.FlatIndex(X.shape[1], metric, 0); idx.set_option("flat_shape", shape); idx.add(X)
D0, I0 = idx.search(Q, k)
assert idx.stats()["scan_shape"] == shape
for variant, name in (((16 + 3, "scan_i8x16_kernel<2,8,8> (16x16x64 MFMA, 1024-query tiles, 8-tile stages)"),) if shape == 16 else
                      ((16 + 3, "scan_i8_kernel<4,8,4> (1024-query tiles, 8-tile stages)"),
                       (16 + 6, "scan_i8_kernel<4,8,4> + a pacing s_barrier per tile (i8_variant 6)"),
                       (16 + 1, "scan_i8_kernel<4,8,2> (512-query tiles)"))):
    idx.set_option("i8_variant", variant)
    for _ in range(25):        # (the clock settles under back-to-back launches)
        D, I = idx.search(Q, k)
    assert np.array_equal(I, I0) and np.array_equal(D, D0), "the stamped build must stay exact"
    buf = np.zeros(4096 * 8 * 8 + 64, np.uint64); n = ctypes.c_int64(0)
    _ffi.check(_ffi.load().vdb_debug_fetch_stamps(idx._h, buf.ctypes.data, buf.size, ctypes.byref(n)))
    w = buf[: n.value].reshape(-1, 8).astype(np.float64)
    w = w[w[:, 5] > 0]
    late = (w[:, 7].astype(np.int64) & 1) == 1
    stages = (w[:, 7].astype(np.int64) >> 1).astype(np.float64)
    cb = 2 if variant == 17 else 4
    print(f"== {name}: {len(w)} waves, in-kernel clock {np.median(w[:,5] / w[:,6]) * 0.1:.3f} GHz (median; p10 {np.percentile(w[:,5]/w[:,6],10)*0.1:.3f}, p90 {np.percentile(w[:,5]/w[:,6],90)*0.1:.3f})")
    for label, sel in (("early half", ~late), ("late half", late)):
        v, st = w[sel], stages[sel]
        tiles = st * 8
        ideal = 4 * cb * 32.0      # MFMA cycles of one tile for one wave: KS x CB instructions of 32 cycles
        per = lambda col: (v[:, col] / tiles).mean()
        print(f"  {label}: cycles per tile per wave: head {per(0):6.0f}  mfma {per(1):6.0f} (pipe time {ideal:.0f})  select {per(2):6.0f}  "
              f"tail {per(3):5.0f}  barrier {per(4):6.0f}  | total {(v[:,5]/tiles).mean():6.0f}  = {2 * ideal / (v[:,5]/tiles).mean():.3f} pipe share for the two waves of a SIMD")
idx.close()
