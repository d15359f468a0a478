#!/usr/bin/env python3
"""Stress of apv_jdiag_leading (csrc/kernels_gevd_lead.hip): pencils of many shapes -- graded, clustered, rank-deficient bright matrices,
tiny and huge scales, ill-conditioned dark matrices -- each checked against LAPACK: eigenvalues, B-orthonormality, residuals,
and which solver answered.  Prints one line per family; exits non-zero on the first violation."""
import os, sys, time
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine


def pencil(lam, rng, cond_b):
    n = lam.size
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    d = np.exp(rng.uniform(-0.5 * np.log(cond_b), 0.5 * np.log(cond_b), n) / 2)
    Xi = Q * d[None, :]
    Bl = Xi @ Xi.T
    A = Xi @ np.diag(lam) @ Xi.T
    return 0.5 * (A + A.T), 0.5 * (Bl + Bl.T) - 1e-7 * np.eye(n)


def families(rng):
    yield "graded 1/i", lambda n: 10.0 / (1 + np.arange(n))
    yield "geometric 0.9^i", lambda n: 0.9 ** np.arange(n)
    yield "linear", lambda n: np.linspace(5.0, 0.01, n)
    yield "pairs (2 % apart)", lambda n: np.repeat(np.linspace(8.0, 0.1, (n + 1) // 2), 2)[:n] * np.tile([1.0, 0.98], (n + 1) // 2)[:n]
    yield "rank 40 bright matrix", lambda n: np.concatenate([np.linspace(3.0, 1.0, 40), np.zeros(n - 40)])
    yield "cluster of 6 at the top", lambda n: np.concatenate([np.full(6, 4.0) * (1 + 1e-9 * np.arange(6)), np.linspace(2.0, 0.05, n - 6)])
    yield "scale 1e-12", lambda n: 1e-12 * np.linspace(5.0, 0.01, n)
    yield "scale 1e+9", lambda n: 1e9 / (1 + np.arange(n)) ** 0.5
    yield "random (noise-like)", lambda n: np.sort(rng.chisquare(8, n))[::-1]


def main():
    rng = np.random.default_rng(2025)
    eng = Engine(1, 4, 4)
    worst = 0.0
    t0 = time.time()
    for name, spec in families(rng):
        stats = []
        for n, rank, cond_b in ((96, 8, 1e2), (200, 16, 1e4), (256, 8, 1e6), (320, 40, 1e3), (416, 50, 1e2)):
            lam = np.sort(np.asarray(spec(n), float))[::-1]
            A, B = pencil(lam, rng, cond_b)
            U, lv, info = eng.jdiag_leading(A[None], B[None], rank)
            U, lv = U[0], lv[0]
            Bl = B + 1e-7 * np.eye(n)
            ref = sl.eigh(A, Bl, eigvals_only=True)[::-1][:rank]
            scale = max(abs(ref[0]), 1e-300)
            e_lam = np.abs(lv - ref).max() / scale
            e_orth = np.abs(U.T @ Bl @ U - np.eye(rank)).max()
            e_res = (np.linalg.norm(A @ U - (Bl @ U) * lv, axis=0) / (np.linalg.norm(A, 2) * np.linalg.norm(U, axis=0))).max()
            stats.append((n, rank, int(info[0]), e_lam, e_orth, e_res))
            worst = max(worst, e_lam, e_res)
            if not (e_lam < 1e-9 and e_orth < 1e-9 and e_res < 1e-9):
                print(f"VIOLATION {name}: n={n} rank={rank} cond(B)={cond_b:g} info={info[0]} lam {e_lam:.2e} orth {e_orth:.2e} res {e_res:.2e}")
                sys.exit(1)
        print(f"{name:28s} " + "  ".join(f"n={n} V={r} {'lead' if i == 0 else 'full'} lam {el:.0e} res {er:.0e}" for n, r, i, el, eo, er in stats), flush=True)
    eng.close()
    print(f"all within 1e-9 (worst {worst:.1e}), {time.time() - t0:.1f} s")


main()
