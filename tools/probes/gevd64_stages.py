"""Time of the order-64 kernel up to each stage (cfg.debug_stop), and the sweeps / refinement steps the bins take."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ap_vast_unofficial_amd import Engine
rng = np.random.default_rng(1234)
def cn(*s):
    return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
XB, XD, d = cn(K, 128, 64), cn(K, 128, 64), cn(K, 128)
MS = int(os.environ.get("MAX_SWEEPS", "0"))
names = {7: "block Jacobi w/o inner solves", 8: "block Jacobi w/o outer update", 1: "correlate", 2: "+cholesky", 3: "+inverse", 4: "+C=WAW^H", 5: "+float block Jacobi", 6: "+refinement", 0: "everything"}
prev = 0.0
for stop in ((4, 5, 7, 8) if MS else (1, 2, 3, 4, 5, 6, 0)):
    eng = Engine(K, 64, 128, ranks=(1, 32, 64), mu=1.0, compute_dtype="f64", debug_stop=stop, max_sweeps=MS)
    dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
    dw = eng.alloc(K * 3 * 64 * 16); dl = eng.alloc(K * 64 * 8); ds = eng.alloc(K * 4)
    for _ in range(2):
        eng.update_dev(dXB, dXD, dd, dw, dl, ds)
    eng.sync()
    eng.timer_start()
    for _ in range(5):
        eng.update_dev(dXB, dXD, dd, dw, dl, ds)
    ms = eng.timer_stop() / 5
    extra = ""
    if stop in (5, 6, 7, 8):
        st = ds.download((K,), np.int32)
        sw, rf = st // 100, st % 100
        extra = f"  sweeps {np.bincount(sw)[sw.min():]} from {sw.min()}" + (f", refinement steps {np.bincount(rf)[rf.min():]} from {rf.min()}" if stop == 6 else "")
    print(f"{names[stop]:22s} {ms:8.3f} ms  (+{ms - prev:6.3f}){extra}", flush=True)
    prev = ms
    eng.close()
