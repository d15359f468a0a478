#!/usr/bin/env python3
"""jdiag_large on the golden broadband pair of cfg1 (n = 256), twice batched, a few repetitions: wall time per
call, and with APV_NO_GRAPH=1 under `rocprofv3 --kernel-trace --stats` the duration of every block round."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ap_vast_unofficial_amd import Engine

g = np.load(os.path.join(ROOT, "tests", "golden", "g1_broadband_cfg1.npz"))
n = 256
def full(t):
    R = np.zeros((n, n)); R[np.triu_indices(n)] = t
    return R + np.triu(R, 1).T
A, B = full(g["R_AA_triu"]), full(g["R_AB_triu"])
A2, B2 = np.stack([A, A]), np.stack([B, B])
eng = Engine(1, 4, 4)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
eng.jdiag_large(A2, B2)
t0 = time.perf_counter()
for _ in range(reps):
    eng.jdiag_large(A2, B2)
print(f"jdiag_large n=256 batch=2: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call (host buffers in and out)")
eng.close()
