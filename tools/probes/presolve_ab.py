#!/usr/bin/env python3
"""A/B of the float32 pre-solve of gevd16m (float64 kernel): debug_stop = 4 runs the double sweeps alone.
Same 32 768 updates as bench.py; prints ms per launch and the worst deviation of the filters between the two."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
import bench
K = 32 * 1024
XB, XD, d = bench.synth(K, 1234)
out = {}
for name, stop in (("mixed", 0), ("double only", 4)):
    eng = Engine(K, 16, 32, ranks=(8,), compute_dtype="f64", out_c128=True, debug_stop=stop)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    dw, ds = eng.alloc(K * 16 * 16), eng.alloc(K * 4)
    for _ in range(20): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    eng.sync(); eng.timer_start()
    for _ in range(100): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    ms = eng.timer_stop() / 100
    out[name] = dw.download((K, 16), np.complex128)
    print(f"{name:12s} {ms:.4f} ms per launch = {K / ms * 1e3:.3e} updates/s, status != 0 in {int((ds.download((K,), np.int32) != 0).sum())} bins")
    eng.close()
e = np.linalg.norm(out["mixed"] - out["double only"], axis=1) / np.linalg.norm(out["double only"], axis=1)
print(f"filters, mixed vs double only: max relative difference {e.max():.2e}, median {np.median(e):.2e}")
