#!/usr/bin/env python3
"""Would the broadband GEVD converge faster when the Jacobi sweeps start from the previous hop's eigenvectors?
CPU model (no GPU): whitened matrices of consecutive hops from the oracle, round-robin cyclic Jacobi in NumPy
(all disjoint pairs of a round rotated at once), sweeps counted until off^2 <= 1e-16 ||C||^2 as the kernel does.
Also counts the sweeps with sorting rotations (the root of the tangent equation that leaves the larger diagonal entry first).
  python tools/probes/warm_start_model.py [cfg1|reftest] [hops]
Result (cfg1, white-noise input, a quarter of the statistics window replaced per hop): the previous eigenvectors leave
off^2/||C||^2 = 0.2-0.3, and the count goes from 11-12 sweeps to 9-10: not worth a hidden state that would make a resumed
stream differ from an uninterrupted one in the last bits.  Sorting rotations with the round-robin order: 19-21 sweeps."""
import os, sys
import numpy as np
import scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.broadband import BroadbandOracle


def whiten(A, B, reg=1e-7):
    L = np.linalg.cholesky(B + reg * np.eye(len(B)))
    C0 = sla.solve_triangular(L, A, lower=True)
    C = sla.solve_triangular(L, C0.T, lower=True).T
    return 0.5 * (C + C.T)


def rr_rounds(n):
    idx = list(range(n))
    for _ in range(n - 1):
        yield [(min(idx[i], idx[n - 1 - i]), max(idx[i], idx[n - 1 - i])) for i in range(n // 2)]
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]


def jacobi_sweeps(C, tol2=1e-16, cap=30, sort=False):
    n = len(C)
    C = C.copy()
    Q = np.eye(n)
    norm2 = (C ** 2).sum()
    hist = []
    for sweep in range(cap):
        piv = 0.0
        for pairs in rr_rounds(n):
            p = np.array([a for a, _ in pairs]); q = np.array([b for _, b in pairs])
            app, aqq, apq = C[p, p], C[q, q], C[p, q]
            piv += 2 * (apq ** 2).sum()
            with np.errstate(divide="ignore", invalid="ignore"):
                tau = (aqq - app) / (2 * apq)
                t = np.sign(tau) / (np.abs(tau) + np.sqrt(1 + tau ** 2))
            t = np.where(apq == 0, 0.0, np.where(np.isfinite(t), t, 0.0))
            t = np.where((tau == 0) & (apq != 0), 1.0, t)
            if sort:                                  # the other root of the tangent equation swaps the diagonal pair
                swap = app < aqq
                with np.errstate(divide="ignore"):
                    tb = np.where(t != 0, -1.0 / np.where(t != 0, t, 1.0), np.inf)
                t = np.where(swap, tb, t)
            c = 1 / np.sqrt(1 + t * t); s = t * c
            if sort:
                c = np.where(np.isinf(t), 0.0, c); s = np.where(np.isinf(t), 1.0, s)
            J = np.eye(n)
            J[p, p] = c; J[q, q] = c; J[p, q] = s; J[q, p] = -s
            C = J.T @ C @ J
            Q = Q @ J
        hist.append(piv / norm2)
        if piv <= tol2 * norm2:
            return sweep + 1, hist, Q, C
    return cap, hist, Q, C


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    hops = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    g = np.load(os.path.join(ROOT, "tests", "golden", "rirs_cfg1.npz"))
    if which == "cfg1":
        N, J, V, S, H, args = 256, 32, 8, 512, 128, (16, 0, 0)
    else:
        N, J, V, S, H, args = 1600, 100, 50, 1000, 800, (20, 6, 6)
        if len(sys.argv) > 3:
            J = int(sys.argv[3])
    np.random.seed(0)
    orc = BroadbandOracle(N, g["rirA"], g["rirB"], J, args[0], args[1], args[2], V, 1.0, S, hop_size=H)
    x = np.random.default_rng(7).standard_normal((2, (hops + 4) * H))
    Qp = None
    for h in range(hops + 4):
        orc.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        if h < 4:
            continue
        C = whiten(orc.R_AA, orc.R_AB)
        cold, hc, Q, D = jacobi_sweeps(C)
        srt, hs, _, _ = jacobi_sweeps(C, sort=True)
        line = f"hop {h}: n={len(C)} cold {cold} sweeps, sorting rotations {srt} ({' '.join('%.0e' % v for v in hs)})"
        if Qp is not None:
            Cw = Qp.T @ C @ Qp
            Cw = 0.5 * (Cw + Cw.T)
            off = ((Cw - np.diag(np.diag(Cw))) ** 2).sum() / (Cw ** 2).sum()
            warm, hw, Q2, _ = jacobi_sweeps(Cw)
            line += f", warm {warm} sweeps (off^2/||C||^2 at start {off:.2e}; per sweep: {' '.join('%.0e' % v for v in hw)})"
            Q = Qp @ Q2
        else:
            line += " (per sweep: " + " ".join("%.0e" % v for v in hc) + ")"
        print(line, flush=True)
        Qp = Q


main()
