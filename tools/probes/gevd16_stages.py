#!/usr/bin/env python3
"""Stage times of the float64 order-16 kernel on the bench workload: the kernel returns early at debug_stop =
1 (correlate), 2 (+ Cholesky / inverse), 3 (+ whitening), 12 (+ float Cholesky), 13 (+ one-sided sweeps), 0 (everything)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
import bench
K = 32 * 1024
XB, XD, d = bench.synth(K, 1234)
prev = 0.0
for name, stop in (("correlate", 1), ("cholesky + inverse", 2), ("whitening", 3), ("float cholesky", 12), ("one-sided sweeps", 13), ("refinement + tail", 0)):
    eng = Engine(K, 16, 32, ranks=(1,), compute_dtype="f64", out_c128=True, debug_stop=stop)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    dw, ds = eng.alloc(K * 16 * 16), eng.alloc(K * 4)
    for _ in range(20): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    eng.sync(); eng.timer_start()
    for _ in range(100): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    ms = eng.timer_stop() / 100
    extra = ""
    if stop == 13:
        st = ds.download((K,), np.int32)
        extra = f"   sweeps: mean {st.mean():.2f}, histogram {dict(zip(*np.unique(st, return_counts=True)))}"
    print(f"{name:20s} cumulative {ms:.4f} ms   stage {ms - prev:.4f} ms{extra}")
    prev = ms
    eng.close()
