#!/usr/bin/env python3
"""test_update_vs_oracle's rank-deficient case (L = 16, M = 8, reg 1e-2) in float32 arithmetic: filter error against the
oracle per solver variant (debug_stop 0 = one-sided, 11 = two-sided sweeps) and seed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
from oracle import subband

def cn(rng, *shape):
    return ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)).astype(np.complex64)

def w_err(a, b):
    return (np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)).max()

K, L, M, ranks = 64, 16, 8, (1, 4, 8)
for seed in (1019, 1, 2):
    rng = np.random.default_rng(seed)
    XB, XD, d = cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M)
    w_ref, lam_ref, _ = subband.update(XB, XD, d, 0.7, list(ranks), reg=1e-2)
    out = []
    for dt, stop in (("f32", 0), ("f32", 11), ("f64", 0), ("f64", 11), ("f64", 5), ("f64", 4)):
        eng = Engine(K, L, M, ranks=ranks, mu=0.7, compute_dtype=dt, reg_dark=1e-2, debug_stop=stop)
        w, lam, status = eng.update(XB, XD, d)
        eng.close()
        e = np.linalg.norm(w - w_ref, axis=-1) / np.linalg.norm(w_ref, axis=-1)
        k, t = np.unravel_index(e.argmax(), e.shape)
        out.append(f"{dt}/{stop}: max {e.max():.2e} (bin {k} rank {ranks[t]}) median {np.median(e):.2e}")
    print(seed, " | ".join(out))
    k = int(np.argmax(np.max(np.linalg.norm(w - w_ref, axis=-1) / np.linalg.norm(w_ref, axis=-1), axis=1)))
    print("   eigenvalues of the worst bin (oracle):", np.array2string(lam_ref[k][:9], precision=4))
