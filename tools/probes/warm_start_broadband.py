#!/usr/bin/env python3
"""Would the broadband joint diagonalisation (n = J L = 256 at cfg1) converge in fewer Jacobi sweeps if every hop started from the
previous hop's generalised eigenvectors?  CPU model: the oracle's (R_bright, R_dark) pairs of consecutive hops (cfg1, white-noise
input, S = 512, H = 128: a quarter of the statistics window is new every hop), a cyclic Jacobi with the round-robin ordering of
the device kernel, sweeps counted until sum |pivot|^2 <= 1e-16 ||C||^2 is met (the device's stop rule), cold (C = W A W^T) against
warm (the same on A' = X_p^T A X_p, B' = X_p^T (B + reg I) X_p)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.broadband import BroadbandOracle


def jacobi_sweeps_rr(C, tol2=1e-16, max_sweeps=30):
    """cyclic Jacobi, tournament ordering (player 0 fixed), vectorised per round"""
    n = C.shape[0]
    C = C.copy()
    norm2 = (C * C).sum()
    players = list(range(n))
    for sweep in range(1, max_sweeps + 1):
        off = 0.0
        for r in range(n - 1):
            a = np.array(players[: n // 2])
            b = np.array(players[n // 2:][::-1])
            p, q = np.minimum(a, b), np.maximum(a, b)
            app, aqq, apq = C[p, p], C[q, q], C[p, q]
            off += float((apq * apq).sum())
            safe = np.where(apq == 0, 1.0, apq)
            tau = (aqq - app) / (2.0 * safe)
            t = np.where(tau >= 0, 1.0, -1.0) / (np.abs(tau) + np.sqrt(1.0 + tau * tau))
            t = np.where(apq == 0, 0.0, t)
            c = 1.0 / np.sqrt(1.0 + t * t)
            s = t * c
            Cp, Cq = C[:, p].copy(), C[:, q].copy()
            C[:, p] = c * Cp - s * Cq
            C[:, q] = s * Cp + c * Cq
            Rp, Rq = C[p, :].copy(), C[q, :].copy()
            C[p, :] = c[:, None] * Rp - s[:, None] * Rq
            C[q, :] = s[:, None] * Rp + c[:, None] * Rq
            players = [players[0]] + [players[-1]] + players[1:-1]
        if off <= tol2 * norm2:
            return sweep
    return max_sweeps


def whiten(A, B):
    Lc = np.linalg.cholesky(B)
    W = np.linalg.inv(Lc)
    C = W @ A @ W.T
    return 0.5 * (C + C.T), W


def main():
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "rirs_cfg1.npz"))
    np.random.seed(0)
    N, H, J, S = 256, 128, 32, 512
    o = BroadbandOracle(N, g["rirA"], g["rirB"], J, 16, 0, 0, 8, 1.0, S, hop_size=H)
    x = np.random.default_rng(7).standard_normal((2, 12 * H))
    Xp = None
    print("| hop | sweeps cold | sweeps warm | off/norm of the warm start |\n|---|---|---|---|")
    for h in range(12):
        o.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        A, B = o.R_AA, o.R_AB + 1e-7 * np.eye(J * 8)
        C, W = whiten(A, B)
        cold = jacobi_sweeps_rr(C)
        warm, rel = None, None
        if Xp is not None:
            C2, _ = whiten(Xp.T @ A @ Xp, Xp.T @ B @ Xp)
            d = np.diag(C2)
            rel = np.sqrt(((C2 * C2).sum() - (d * d).sum()) / (C2 * C2).sum())
            warm = jacobi_sweeps_rr(C2)
        lam, Q = np.linalg.eigh(C)
        Xp = W.T @ Q[:, ::-1]
        print(f"| {h} | {cold} | {warm} | {rel if rel is None else f'{rel:.3f}'} |", flush=True)


main()
