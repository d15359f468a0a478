"""cfg5 at full size (2048 bins) with different pre-solve hand-over thresholds (cfg.sweep_tol2 < 0 = tuning aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ap_vast_unofficial_amd import Engine
rng = np.random.default_rng(1234)
def cn(*s):
    out = np.empty(s, np.complex64)
    out.real = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(.5))
    out.imag = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(.5))
    return out
K = 2048
XB, XD, d = cn(K, 128, 64), cn(K, 128, 64), cn(K, 128)
for tol in (0.0, -1e-10, -1e-8, -1e-7, -1e-6, -1e-5):
    for stop in (6, 0):
        eng = Engine(K, 64, 128, ranks=(1, 32, 64), compute_dtype="f64", out_c128=False, sweep_tol2=tol, debug_stop=stop)
        dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
        dw, ds = eng.alloc(K * 3 * 64 * 8), eng.alloc(K * 4)
        eng.update_dev(dXB, dXD, dd, dw, None, ds); eng.sync()
        eng.timer_start()
        for _ in range(3): eng.update_dev(dXB, dXD, dd, dw, None, ds)
        ms = eng.timer_stop() / 3
        st = ds.download((K,), np.int32)
        if stop == 6:
            print(f"hand-over {-tol if tol else 'default':>8}: sweeps {np.bincount(st // 100)[5:]} (from 5), refinement steps {np.bincount(st % 100)[1:]} (from 1)", end="  ")
        else:
            print(f"{ms:.3f} ms / 2048 bins = {K / ms * 1e3:.3e} updates/s, status != 0 in {(st != 0).sum()}", flush=True)
        eng.close()
