#!/usr/bin/env python3
"""Sweeps of the float pre-solve per bin on the bench workload (debug_stop = 13 leaves the count in the status word), and how
many bins take a second refinement step (debug_stop = 9 marks them with status 8)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
import bench
K = 32 * 1024
XB, XD, d = bench.synth(K, 1234)
for name, stop in (("sweeps of the pre-solve", 13), ("second refinement step (status 8)", 9)):
    eng = Engine(K, 16, 32, ranks=(8,), compute_dtype="f64", out_c128=True, debug_stop=stop)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    dw, ds = eng.alloc(K * 16 * 16), eng.alloc(K * 4)
    eng.update_dev(dXB, dXD, dd, dw, None, ds)
    eng.sync()
    st = ds.download((K,), np.int32)
    vals, cnt = np.unique(st, return_counts=True)
    print(f"{name}: " + ", ".join(f"{v}: {c} ({100.0 * c / K:.1f} %)" for v, c in zip(vals, cnt)) + (f"; mean {st.mean():.3f}" if stop == 13 else ""))
    eng.close()
