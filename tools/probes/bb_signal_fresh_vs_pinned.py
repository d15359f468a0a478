#!/usr/bin/env python3
"""process_signal in broadband mode at cfg1: a fresh result array per call against the caller's page-locked array, alternating."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ap_vast_unofficial_amd.apvast import apvast

g = np.load(os.path.join(ROOT, "tests", "golden", "rirs_cfg1.npz"))
ref = len(sys.argv) > 1 and sys.argv[1] == "ref"
if ref:
    N, J, V, S, H, hops = 1600, 100, 50, 1000, 800, 16
    ap = apvast(N, g["rirA"], g["rirB"], J, 20, 6, 6, V, 1.0, S, perceptual=False, mode="broadband", seed=0)
else:
    N, H, hops = 256, 128, 128
    ap = apvast(N, g["rirA"], g["rirB"], 32, 16, 0, 0, 8, 1.0, 512, hop_size=H, perceptual=False, mode="broadband", seed=0)
xs = np.random.default_rng(8).standard_normal((2, hops * H))
ap.process_signal(xs[0, :16 * H], xs[1, :16 * H])
out = ap.alloc_signal_output(hops * H)
out[...] = 0
for rep in range(3):
    t0 = time.perf_counter(); r = ap.process_signal(xs[0], xs[1]); t1 = time.perf_counter(); del r
    t2 = time.perf_counter(); ap.process_signal(xs[0], xs[1], out=out); t3 = time.perf_counter()
    print("fresh %.4f ms/hop   page-locked out %.4f ms/hop" % ((t1 - t0) / hops * 1e3, (t3 - t2) / hops * 1e3))
ap.close()
