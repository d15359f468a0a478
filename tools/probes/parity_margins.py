#!/usr/bin/env python3
"""Measured errors behind the bounds of tests/test_gpu_parity.py::test_update_vs_oracle: every case of the test, the three error
figures it asserts on, next to the SURVEY 8(c) tolerance."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
from oracle import subband

TOL = {"f64": dict(lam=1e-9, w=1e-7), "f32": dict(lam=1e-5, w=1e-4)}
CASES = [(37, 16, 32, (1, 8, 16)), (5, 10, 24, (1, 5, 10)), (9, 5, 7, (2, 5)), (3, 32, 48, (1, 16, 32)), (2, 64, 128, (1, 32, 64)),
         (4, 1, 3, (1,)), (3, 16, 8, (1, 4, 8))]


def cn(rng, *s):
    return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)


print("| dtype | K, L, M | lam abs/lam_1 (leading min(L, M)) | lam rel (all, M >= L) | w rel | SURVEY lam / w |\n|---|---|---|---|---|---|")
for dtype in ("f64", "f32"):
    for K, L, M, ranks in CASES:
        rng = np.random.default_rng(1000 + K + L)
        XB, XD, d = cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M)
        reg = 1e-7 if M >= L else 1e-2
        eng = Engine(K, L, M, ranks=ranks, mu=0.7, compute_dtype=dtype, reg_dark=reg)
        w, lam, status = eng.update(XB, XD, d)
        eng.close()
        w_ref, lam_ref, _ = subband.update(XB, XD, d, 0.7, list(ranks), reg=reg)
        nz = min(L, M)
        e1 = (np.abs(lam[:, :nz] - lam_ref[:, :nz]) / lam_ref[:, :1]).max()
        e2 = np.abs(lam / lam_ref - 1).max() if M >= L else float("nan")
        good = [t for t, V in enumerate(ranks) if V <= nz]
        e3 = (np.linalg.norm(w[:, good] - w_ref[:, good], axis=-1) / np.linalg.norm(w_ref[:, good], axis=-1)).max()
        print(f"| {dtype} | {K}, {L}, {M} | {e1:.2e} | {e2:.2e} | {e3:.2e} | {TOL[dtype]['lam']:.0e} / {TOL[dtype]['w']:.0e} |")
