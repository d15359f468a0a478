"""Order-64 kernel against the oracle: golden G3 64x128 and random cfg5-shaped bins; time against the LDS kernel
(APV_NO_GEVD64=1 in a second process selects it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ap_vast_unofficial_amd import Engine
from oracle import subband

g = np.load(os.path.join(ROOT, "tests/golden/g3_jdiag_c_64x128.npz"))
XB, XD, d = g["XB"], g["XD"], g["d"]
K, M, L = XB.shape
ranks = [int(v) for v in g["ranks"]]
eng = Engine(K, L, M, ranks=ranks, mu=float(g["mu"]), compute_dtype="f64", reg_dark=float(g["reg"]))
w, lam, status = eng.update(XB, XD, d, raise_on_status=False)
eng.close()
werr = np.linalg.norm(w - g["w"], axis=-1) / np.linalg.norm(g["w"], axis=-1)
print(f"G3 64x128: status {np.unique(status)}, lam rel err {np.abs(lam / g['lam'] - 1).max():.2e}, w err {werr.max():.2e}", flush=True)

rng = np.random.default_rng(1234)
def cn(*s):
    return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)
K = 256
XB, XD, d = cn(K, 128, 64), cn(K, 128, 64), cn(K, 128)
eng = Engine(K, 64, 128, ranks=(1, 32, 64), mu=1.0, compute_dtype="f64")
w, lam, status = eng.update(XB, XD, d, raise_on_status=False)
wr, lr = subband.update_vectorised(XB, XD, d, 1.0, [1, 32, 64])
werr = np.linalg.norm(w - wr, axis=-1) / np.linalg.norm(wr, axis=-1)
print(f"random {K} bins: status {np.unique(status, return_counts=True)}, lam rel err {np.abs(lam / lr - 1).max():.2e}, w err max {werr.max():.2e} median {np.median(werr):.2e}", flush=True)
dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
dw = eng.alloc(K * 3 * 64 * 16); dl = eng.alloc(K * 64 * 8); ds = eng.alloc(K * 4)
for _ in range(3):
    eng.update_dev(dXB, dXD, dd, dw, dl, ds)
eng.sync()
eng.timer_start()
for _ in range(10):
    eng.update_dev(dXB, dXD, dd, dw, dl, ds)
ms = eng.timer_stop() / 10
print(f"{K} bins: {ms:.3f} ms per launch = {K / ms * 1e3:.3e} updates/s ({'LDS kernel' if os.environ.get('APV_NO_GEVD64') else 'order-64 kernel'})", flush=True)
eng.close()
