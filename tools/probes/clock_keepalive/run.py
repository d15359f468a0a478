#!/usr/bin/env python3
"""Per-hop broadband calls with and without a wave that keeps the device busy beside them (tools/probes/clock_keepalive/spin.hip)."""
import ctypes, os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
from ap_vast_unofficial_amd.apvast import apvast
spin = ctypes.CDLL(os.path.join(HERE, "libspin.so"))
spin.spin_start.argtypes = [ctypes.c_double, ctypes.c_int]
g = np.load(os.path.join(ROOT, "tests", "golden", "rirs_cfg1.npz"))
for name, mk, H, hops in (("cfg1", lambda: apvast(256, g["rirA"], g["rirB"], 32, 16, 0, 0, 8, 1.0, 512, hop_size=128, perceptual=False, mode="broadband", seed=0), 128, 60),
                          ("n=800", lambda: apvast(1600, g["rirA"], g["rirB"], 100, 20, 6, 6, 50, 1.0, 1000, perceptual=False, mode="broadband", seed=0), 800, 12)):
    ap = mk()
    x = np.random.default_rng(7).standard_normal((2, (hops + 2) * H))
    for h in range(2):
        ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    for waves in (0, 1, 64):
        if waves:
            assert spin.spin_start(4000.0, waves) == 0
            time.sleep(0.05)
        t0 = time.perf_counter()
        for h in range(2, hops + 2):
            ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        dt = (time.perf_counter() - t0) / hops
        print("%s: %d busy waves beside the hops: %.3f ms/hop" % (name, waves, dt * 1e3))
        if waves:
            spin.spin_wait()
    ap.close()
