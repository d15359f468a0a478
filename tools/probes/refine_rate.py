#!/usr/bin/env python3
"""How many bins of the bench workload fail the refinement guard of gevd16m (debug_stop = 9 marks them with status 8),
and what the guarded path costs (debug_stop = 10 switches the guard off: timing only, results invalid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
import bench
K = 32 * 1024
XB, XD, d = bench.synth(K, 1234)
for name, stop in (("normal", 0), ("mark", 9), ("guard off", 10), ("two-sided pre-solve", 11)):
    eng = Engine(K, 16, 32, ranks=(1,), compute_dtype="f64", out_c128=True, debug_stop=stop)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    dw, ds = eng.alloc(K * 16 * 16), eng.alloc(K * 4)
    for _ in range(20): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    eng.sync(); eng.timer_start()
    for _ in range(100): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    ms = eng.timer_stop() / 100
    st = ds.download((K,), np.int32)
    print(f"{name:10s} {ms:.4f} ms per launch; status counts {dict(zip(*np.unique(st, return_counts=True)))}")
    eng.close()
