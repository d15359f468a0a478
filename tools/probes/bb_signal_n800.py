#!/usr/bin/env python3
"""process_signal at the reference's test parameters (n = 800): one warm call of 16 hops, wall time per hop.  Run under
rocprofv3 --kernel-trace --stats for the per-kernel split, or with APV_LEAD_DEBUG=1 for the passes of the leading solver."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ap_vast_unofficial_amd.apvast import apvast

g = np.load(os.path.join(ROOT, "tests", "golden", "rirs_cfg1.npz"))
N, J, V, S, H = 1600, 100, 50, 1000, 800
hops = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ap = apvast(N, g["rirA"], g["rirB"], J, 20, 6, 6, V, 1.0, S, perceptual=False, mode="broadband", seed=0)
xs = np.random.default_rng(8).standard_normal((2, (hops + 16) * H))
ap.process_signal(xs[0, :16 * H], xs[1, :16 * H])
sys.stderr.write("---- timed call ----\n")
t0 = time.perf_counter()
ap.process_signal(xs[0, 16 * H:], xs[1, 16 * H:])
dt = time.perf_counter() - t0
print("process_signal n=800: %.3f ms/hop over %d hops, not converged %d" % (dt / hops * 1e3, hops, ap.not_converged))
t0 = time.perf_counter()
out = ap.alloc_signal_output(hops * H)
t_alloc = time.perf_counter() - t0
for rep in range(2):
    t0 = time.perf_counter()
    ap.process_signal(xs[0, 16 * H:], xs[1, 16 * H:], out=out)
    dt = time.perf_counter() - t0
    print("  into a page-locked array (allocated in %.2f ms): %.3f ms/hop" % (t_alloc * 1e3, dt / hops * 1e3))
t0 = time.perf_counter()
for h in range(6):
    ap.process_input_buffers(xs[0, h * H:(h + 1) * H], xs[1, h * H:(h + 1) * H])
print("  per-hop calls: %.3f ms/hop" % ((time.perf_counter() - t0) / 6 * 1e3))
ap.close()
