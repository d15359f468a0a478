#!/usr/bin/env python3
"""Eigenvalue error of the fused order-16 update on fixture G3 per pre-solve variant (debug_stop 0 / 11) and which bins
took a second refinement step (debug_stop 9)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "g3_jdiag_c_16x32.npz"))
XB, XD, d = g["XB"], g["XD"], g["d"]
K, M, L = XB.shape
ranks = [int(v) for v in g["ranks"]]
res = {}
for stop in (0, 11, 9, 5):
    eng = Engine(K, L, M, ranks=ranks, mu=float(g["mu"]), compute_dtype="f64", reg_dark=float(g["reg"]), debug_stop=stop)
    w, lam, status = eng.update(XB, XD, d)
    eng.close()
    err = np.abs(lam / g["lam"] - 1)
    k, i = np.unravel_index(err.argmax(), err.shape)
    print(f"debug_stop {stop:2d}: max rel lam err {err.max():.3e} at bin {k} index {i} (lam {g['lam'][k, i]:.4g}, lam_max {g['lam'][k, 0]:.4g}); "
          f"bins > 3e-10: {(err.max(1) > 3e-10).sum()}; status {dict(zip(*np.unique(status, return_counts=True)))}")
    res[stop] = (err, status)
err, _ = res[0]
_, st = res[9]
for k in np.argsort(-err.max(1))[:6]:
    gaps = np.diff(g["lam"][k][::-1])
    print(f"  bin {k}: err {err[k].max():.2e} idx {err[k].argmax()} status9 {st[k]} min gap/lam_max {gaps.min() / g['lam'][k, 0]:.2e}")
