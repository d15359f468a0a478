#!/usr/bin/env python3
"""Is the correlate stage of the order-16 kernel (debug_stop = 1: return after it) bound by its work or by workgroup dispatch?
Same 32 768 bins with M = 32, 16, 8 control points (the loads and MFMAs of the stage scale with M)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
K, L = 32 * 1024, 16
rng = np.random.default_rng(0)
for M in (32, 16, 8):
    XB = (rng.standard_normal((K, M, L)) + 1j * rng.standard_normal((K, M, L))).astype(np.complex64)
    XD = (rng.standard_normal((K, M, L)) + 1j * rng.standard_normal((K, M, L))).astype(np.complex64)
    d = (rng.standard_normal((K, M)) + 1j * rng.standard_normal((K, M))).astype(np.complex64)
    for stop in (1, 0):
        eng = Engine(K, L, M, ranks=(8,), compute_dtype="f64", out_c128=True, debug_stop=stop, reg_dark=1e-2)
        dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
        dw, ds = eng.alloc(K * 16 * 16), eng.alloc(K * 4)
        for _ in range(20): eng.update_dev(dXB, dXD, dd, dw, None, ds)
        eng.sync(); eng.timer_start()
        for _ in range(100): eng.update_dev(dXB, dXD, dd, dw, None, ds)
        ms = eng.timer_stop() / 100
        print(f"M = {M:2d}, debug_stop {stop}: {ms:.4f} ms per launch ({(2 * K * M * L * 8 + K * M * 8) / ms / 1e9:.2f} TB/s of input)")
        eng.close()
