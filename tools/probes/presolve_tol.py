#!/usr/bin/env python3
"""Stop threshold of the float pre-solve of the float64 order-16 kernel (debug_stop = 20 + e sets it to 1e-e) against time per
launch; debug_stop 9 at the default shows how many bins need a second refinement step."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
import bench
K = 32 * 1024
XB, XD, d = bench.synth(K, 1234)
ref = None
STOPS = (0, 26, 25, 24, 23, 22) if len(sys.argv) < 2 else (26, 31, 32, 33, 35, 37, 39, 26)     # any argument: the fine scan, 0.5e-6 (stop - 29)
for stop in STOPS:
    eng = Engine(K, 16, 32, ranks=(8,), compute_dtype="f64", out_c128=True, debug_stop=stop)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    dw, ds = eng.alloc(K * 16 * 16), eng.alloc(K * 4)
    for _ in range(20): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    eng.sync(); eng.timer_start()
    for _ in range(100): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    ms = eng.timer_stop() / 100
    w = dw.download((K, 16), np.complex128)
    if ref is None: ref = w
    e = np.linalg.norm(w - ref, axis=1) / np.linalg.norm(ref, axis=1)
    print(f"debug_stop {stop:2d} (tol2 = {'default 1e-6' if stop == 0 else ('1e-%d' % (stop - 20) if stop < 30 else '%.1e' % (0.5e-6 * (stop - 29)))}): {ms:.4f} ms per launch, filters vs default: max {e.max():.1e}")
    eng.close()
