#!/usr/bin/env python3
"""NumPy model of the leading-eigenpair solver of csrc/kernels_gevd_lead.hip, on the oracle's own broadband pairs.

The broadband hop consumes the leading V generalised eigenpairs of (R_bright, R_dark + reg I) (apvast.py:406-414).  This model
was used before any device code to decide (a) whether a block method converges fast enough on the reference's spectra,
(b) the block width, (c) the degree caps of the Chebyshev filter, (d) whether a warm start from the previous hop pays.

    python tools/probes/lead_model.py cfg1      # n = 256, V = 8, block 32  (BASELINE config 1)
    python tools/probes/lead_model.py ref       # n = 800, V = 50, block 64 (make_python_test.m:6-15)

Per pair it prints lambda_{b+1} / lambda_V (what a block of width b has to separate), lambda_1 / lambda_V (the amplification
disparity a filter creates), the Rayleigh-Ritz passes and block products until max_j<V ||C x_j - theta_j x_j|| <= 2e-14 sqrt(n)
lambda_1, the error of the leading-V projector against LAPACK, and -- with `warm` -- the same from the previous hop's block.
Findings (profiles/r04/lead_model.txt): cfg1 3 passes / 15-31 products, n = 800 5 passes / 25-38 products (71 on the first hop,
whose start buffers are noise); projector error <= 1e-11; degree caps 1e6 / 1e10 / 1e12 on T_m(x(theta_1)) keep the Gram matrix
factorisable (uncapped degree 14-16 with lambda_1 / lambda_V = 8: cond 1e16, Cholesky breaks down).  Warm start from the
previous hop's block: at cfg1 (75 % of the statistics window shared) it saves a pass where the smallest cosine between the
old block and the new leading vectors is above 0.8 -- three of fourteen pairs, 2 passes / 12-14 products instead of 3 / 18-21 --
and nothing elsewhere (cosine 0.15-0.75); at n = 800 (20 % shared, cosine 0.005-0.14) nothing at all.  The device code
therefore starts every solve from the same pseudo-random block: stateless, and a resumed stream continues bit for bit."""
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.broadband import BroadbandOracle  # noqa: E402  (a probe, not product code)


def rayleigh_ritz(Y, CY):
    """(H, G) -> T, theta as the device does it: unit diagonal, Cholesky of G, M = L^-1 H L^-T, eigh, T = D L^-T Q."""
    G = Y.T @ Y
    H = Y.T @ CY
    H = 0.5 * (H + H.T)
    d = 1 / np.sqrt(np.diag(G))
    G, H = G * d[:, None] * d[None, :], H * d[:, None] * d[None, :]
    L = np.linalg.cholesky(G)
    Li = sl.solve_triangular(L, np.eye(L.shape[0]), lower=True)
    M = Li @ H @ Li.T
    th, Q = np.linalg.eigh(0.5 * (M + M.T))
    return th[::-1], d[:, None] * (Li.T @ Q[:, ::-1]), np.linalg.cond(G)


def cheb(C, X, CX, m, c, s1):
    """scaled Chebyshev filter of degree m that damps [0, c] (Zhou & Saad 2007); step 1 from the known product C X"""
    e = ctr = 0.5 * c
    sigma1 = e / (s1 - ctr)
    sigma = sigma1
    Y, Xp = (CX - ctr * X) * (sigma1 / e), X
    for _ in range(2, m + 1):
        sn = 1 / (2 / sigma1 - sigma)
        Y, Xp = (C @ Y - ctr * Y) * (2 * sn / e) - (sigma * sn) * Xp, Y
        sigma = sn
    return Y


def solve(C, V, b, X0=None, limits=(1e6, 1e10, 1e12), mmax=16, maxpass=20):
    n = C.shape[0]
    X = np.random.default_rng(1).standard_normal((n, b)) if X0 is None else X0
    CX = C @ X
    th, T, cond = rayleigh_ritz(X, CX)
    X, CX = X @ T, CX @ T
    products = 1
    for it in range(maxpass):
        res = np.linalg.norm(CX - X * th, axis=0)
        if res[:V].max() <= 2e-14 * np.sqrt(n) * th[0]:
            break
        c = max(th[b - 1], 1e-12 * th[0])
        x1 = 2 * th[0] / c - 1
        m = int(np.clip(np.floor(np.arccosh(limits[min(it, len(limits) - 1)]) / np.arccosh(max(x1, 1 + 1e-12))), 1, mmax))
        Y = cheb(C, X, CX, m, c, th[0])
        CY = C @ Y
        products += m
        th, T, cond = rayleigh_ritz(Y, CY)
        X, CX = Y @ T, CY @ T
    return th, X, it, products


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    warm = "warm" in sys.argv
    g = np.load(os.path.join(ROOT, "tests", "golden", "rirs_cfg1.npz"))
    N, J, V, S, H, delay, ref, hops, b = (256, 32, 8, 512, 128, 16, 0, 8, 32) if tag == "cfg1" else (1600, 100, 50, 1000, 800, 20, 6, 4, 64)
    np.random.seed(0)
    o = BroadbandOracle(N, g["rirA"], g["rirB"], J, delay, ref, ref, V, 1.0, S, hop_size=H)
    x = np.random.default_rng(7).standard_normal((2, hops * H))
    prev = [None, None]
    for h in range(hops):
        o.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        for z, (A, B) in enumerate(((o.R_AA, o.R_AB), (o.R_BB, o.R_BA))):
            n = A.shape[0]
            W = sl.solve_triangular(np.linalg.cholesky(B + 1e-7 * np.eye(n)), np.eye(n), lower=True)
            C = W @ A @ W.T
            C = 0.5 * (C + C.T)
            lam, U = np.linalg.eigh(C)
            lam, U = lam[::-1], U[:, ::-1]
            th, X, passes, products = solve(C, V, b)
            P = U[:, :V]
            pe = np.linalg.norm(X[:, :V] - P @ (P.T @ X[:, :V]), 2)
            line = (f"{tag} hop {h} zone {z}: lam_b+1/lam_V {lam[b] / lam[V - 1]:.3f} lam_1/lam_V {lam[0] / lam[V - 1]:.2f} -> {passes} passes, "
                    f"{products} products, projector error {pe:.1e}, lambda error {np.abs(th[:V] - lam[:V]).max() / lam[0]:.1e}")
            if warm and prev[z] is not None:
                # the previous hop's block, expressed in this hop's whitened coordinates (x = W^-T u), re-orthonormalised
                X0, _ = np.linalg.qr(np.linalg.solve(W.T, prev[z]))
                cs = np.linalg.svd(X0.T @ U[:, :V], compute_uv=False).min()
                _, _, pw, prw = solve(C, V, b, X0=X0)
                line += f"; warm start (min cosine {cs:.3f}): {pw} passes, {prw} products"
            print(line, flush=True)
            prev[z] = W.T @ X           # generalised eigenvectors u = W^T x of this hop


if __name__ == "__main__":
    main()
