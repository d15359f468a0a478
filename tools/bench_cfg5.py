#!/usr/bin/env python3
"""cfg5 of BASELINE.json: 64 loudspeakers x 128 control points x 2048 bins (kernel level)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ap_vast_unofficial_amd import Engine

def main():
    L, M, K = 64, 128, 2048
    rng = np.random.default_rng(1234)
    def cn(*s):
        out = np.empty(s, np.complex64)
        out.real = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(.5))
        out.imag = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(.5))
        return out
    XB, XD, d = cn(K, M, L), cn(K, M, L), cn(K, M)
    res = {}
    for dt in sys.argv[1:] or ["f32", "f64"]:
        eng = Engine(K, L, M, ranks=(1, 32, 64), compute_dtype=dt, out_c128=False)
        dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
        dw, ds = eng.alloc(K * 3 * L * 8), eng.alloc(K * 4)
        for _ in range(6): eng.update_dev(dXB, dXD, dd, dw, None, ds)        # clocks ramp over the first launches after idling
        eng.sync()
        eng.timer_start()
        for _ in range(10): eng.update_dev(dXB, dXD, dd, dw, None, ds)
        ms = eng.timer_stop() / 10
        st = ds.download((K,), np.int32)
        bytes_per = 2 * M * L * 8 + M * 8 + 3 * L * 8
        res[dt] = {"ms": ms, "updates_per_s": K / ms * 1e3, "GBps_algorithmic": bytes_per * K / ms / 1e6, "status_nonzero": int((st != 0).sum())}
        eng.close()
    print(json.dumps({"workload": "cfg5 64x128x2048", **res}))
main()
