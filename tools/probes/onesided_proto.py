#!/usr/bin/env python3
"""NumPy model of a float32 ONE-SIDED (Hestenes) Jacobi pre-solve for the order-16 kernel: G = chol(C + delta I) in
float32, column rotations until the columns are orthogonal (G J = U Sigma, so U = eigenvectors of C = G G^H), against the
two-sided float32 Jacobi the kernel runs today.  Batched over bins; counts sweeps and the quality of V32 as the
Ogita-Aishima step sees it (max |Z_ij|)."""
import sys
import numpy as np

rng = np.random.default_rng(0)


def make_C(K, L=16, M=32, reg=1e-7, kind="bench"):
    if kind == "bench":
        XB = (rng.standard_normal((K, M, L)) + 1j * rng.standard_normal((K, M, L))) / np.sqrt(2)
        XD = (rng.standard_normal((K, M, L)) + 1j * rng.standard_normal((K, M, L))) / np.sqrt(2)
    else:  # "stream": decaying spectra, strongly coloured
        sc = np.exp(-np.arange(L) / 2.0)[None, None, :]
        mix = rng.standard_normal((K, L, L)) + 1j * rng.standard_normal((K, L, L))
        XB = ((rng.standard_normal((K, M, L)) + 1j * rng.standard_normal((K, M, L))) * sc) @ mix * 1e-3
        XD = ((rng.standard_normal((K, M, L)) + 1j * rng.standard_normal((K, M, L))) * sc[..., ::-1]) @ mix * 1e-3
    RB = XB.conj().transpose(0, 2, 1) @ XB
    RD = XD.conj().transpose(0, 2, 1) @ XD + reg * np.eye(L)
    Lc = np.linalg.cholesky(RD)
    W = np.linalg.inv(Lc)
    C = W @ RB @ W.conj().transpose(0, 2, 1)
    C = 0.5 * (C + C.conj().transpose(0, 2, 1))
    return C


def round_pairs(sweep):
    gray3 = [0, 1, 3, 2, 6, 7, 5, 4]
    if sweep % 2 == 0:
        rs = [8 ^ g for g in gray3] + [4 ^ g for g in [0, 1, 3, 2]] + [2, 3] + [1]
    else:
        rs = [1 ^ (g << 1) for g in gray3] + [2 ^ (g << 2) for g in [0, 1, 3, 2]] + [4, 12] + [8]
    out = []
    for r in rs:
        p = np.array([i for i in range(16) if i < (i ^ r)])
        out.append((p, p ^ r))
    return out


def angle32(alpha, gamma, beta):
    """(c, s) as rotation() of gevd16_common.h, float32"""
    b2 = (beta.real ** 2 + beta.imag ** 2).astype(np.float32)
    ok = b2 > 1e-30
    iab = np.where(ok, 1.0 / np.sqrt(np.where(ok, b2, 1)), 0).astype(np.float32)
    tau = ((gamma - alpha) * np.float32(0.5) * iab).astype(np.float32)
    rho = np.sqrt(tau * tau + 1).astype(np.float32)
    t = (np.copysign(1.0 / (np.abs(tau) + rho), tau) * iab).astype(np.float32)
    tt = (beta * t).astype(np.complex64)
    c = (1.0 / np.sqrt(1 + tt.real ** 2 + tt.imag ** 2)).astype(np.float32)
    return c, (tt * c).astype(np.complex64)


def two_sided(C, tol2=1e-6, max_sweeps=12):
    K = C.shape[0]
    nf = np.sqrt((np.abs(C) ** 2).sum((1, 2)))
    A = (C / nf[:, None, None]).astype(np.complex64)
    V = np.broadcast_to(np.eye(16, dtype=np.complex64), A.shape).copy()
    sweeps = np.zeros(K, int)
    active = np.ones(K, bool)
    for sw in range(max_sweeps):
        off = np.zeros(K, np.float32)
        for p, q in round_pairs(sw):
            beta = A[:, p, q]
            off += (np.abs(beta) ** 2).sum(1)
            c, s = angle32(A[:, p, p].real, A[:, q, q].real, beta)
            c = np.where(active[:, None], c, 1).astype(np.float32)
            s = np.where(active[:, None], s, 0).astype(np.complex64)
            for Mx in (A, V):       # columns: p' = c p - conj(s) q, q' = c q + s p
                P, Q = Mx[:, :, p].copy(), Mx[:, :, q].copy()
                Mx[:, :, p] = c[:, None, :] * P - np.conj(s)[:, None, :] * Q
                Mx[:, :, q] = c[:, None, :] * Q + s[:, None, :] * P
            P, Q = A[:, p, :].copy(), A[:, q, :].copy()   # rows: p' = c p - s q, q' = c q + conj(s) p
            A[:, p, :] = c[:, :, None] * P - s[:, :, None] * Q
            A[:, q, :] = c[:, :, None] * Q + np.conj(s)[:, :, None] * P
        sweeps += active
        active &= ~(off <= tol2)
        if not active.any():
            break
    return V, sweeps


def one_sided(C, tol2=1e-6, max_sweeps=12, delta=1e-6, chol64=False):
    K = C.shape[0]
    nf = np.sqrt((np.abs(C) ** 2).sum((1, 2)))
    A = C / nf[:, None, None]
    if chol64:
        G = np.linalg.cholesky(A + delta * np.eye(16)).astype(np.complex64)
    else:
        # float32 Cholesky (numpy has no complex64 cholesky distinct from LAPACK's: emulate by rounding the input)
        G = np.linalg.cholesky((A.astype(np.complex64) + np.float32(delta) * np.eye(16, dtype=np.complex64))).astype(np.complex64)
    sweeps = np.zeros(K, int)
    active = np.ones(K, bool)
    for sw in range(max_sweeps):
        off = np.zeros(K, np.float32)
        for p, q in round_pairs(sw):
            P, Q = G[:, :, p].copy(), G[:, :, q].copy()
            beta = (np.conj(P) * Q).sum(1).astype(np.complex64)
            alpha = (np.abs(P) ** 2).sum(1).astype(np.float32)
            gamma = (np.abs(Q) ** 2).sum(1).astype(np.float32)
            off += (np.abs(beta) ** 2).sum(1)
            c, s = angle32(alpha, gamma, beta)
            c = np.where(active[:, None], c, 1).astype(np.float32)
            s = np.where(active[:, None], s, 0).astype(np.complex64)
            G[:, :, p] = c[:, None, :] * P - np.conj(s)[:, None, :] * Q
            G[:, :, q] = c[:, None, :] * Q + s[:, None, :] * P
        sweeps += active
        active &= ~(off <= tol2)
        if not active.any():
            break
    nrm = np.sqrt((np.abs(G) ** 2).sum(1)).astype(np.float32)
    return (G / nrm[:, None, :]).astype(np.complex64), sweeps


def oa_quality(C, V32):
    """max |Z_ij| of the refinement step and the error of the refined eigenvalues / invariance after one step"""
    V = V32.astype(np.complex128)
    VH = V.conj().transpose(0, 2, 1)
    S = VH @ C @ V
    Gm = VH @ V
    d = (np.diagonal(S, axis1=1, axis2=2) / np.diagonal(Gm, axis1=1, axis2=2)).real
    E = Gm - np.eye(16)
    den = d[:, None, :] - d[:, :, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        Z = (S - d[:, None, :] * E) / den
    idx = np.arange(16)
    Z[:, idx, idx] = -0.5 * E[:, idx, idx]
    zoff = Z.copy(); zoff[:, idx, idx] = 0
    zmax = np.abs(zoff).max((1, 2))
    zmax = np.where(np.isfinite(zmax), zmax, np.inf)
    V2 = V + V @ Z
    # residual of refined vectors
    lam = np.linalg.eigvalsh(C)
    R = V2.conj().transpose(0, 2, 1) @ C @ V2
    offd = R.copy(); offd[:, idx, idx] = 0
    res = np.abs(offd).max((1, 2)) / lam.max(1)
    return zmax, res


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    for kind in ("bench", "stream"):
        C = make_C(K, kind=kind)
        lam = np.linalg.eigvalsh(C)
        print(f"== {kind}: cond(C) median {np.median(lam[:, -1] / lam[:, 0]):.2e}, max {np.max(lam[:, -1] / lam[:, 0]):.2e}")
        V2s, s2 = two_sided(C)
        z2, r2 = oa_quality(C, V2s)
        print(f"two-sided f32 : sweeps mean {s2.mean():.2f} max {s2.max()}  |Z|max median {np.median(z2):.2e} pass(3e-5) {np.mean(z2 <= 3e-5):.3f}  resid after OA median {np.median(r2):.1e} max {r2.max():.1e}")
        for delta, c64 in ((1e-6, False), (1e-5, False), (1e-6, True), (1e-8, True)):
            V1s, s1 = one_sided(C, delta=delta, chol64=c64)
            z1, r1 = oa_quality(C, V1s)
            print(f"one-sided d={delta:g} chol64={int(c64)}: sweeps mean {s1.mean():.2f} max {s1.max()}  |Z|max median {np.median(z1):.2e} pass {np.mean(z1 <= 3e-5):.3f}  resid after OA median {np.median(r1):.1e} max {r1.max():.1e}")


def oa_two_steps(C, V32):
    """fraction of bins per outcome with a second refinement step in the rotated basis"""
    V = V32.astype(np.complex128)
    idx = np.arange(16)
    I = np.eye(16)

    def step(S, Gm):
        d = (np.diagonal(S, axis1=1, axis2=2) / np.diagonal(Gm, axis1=1, axis2=2)).real
        E = Gm - I
        with np.errstate(divide="ignore", invalid="ignore"):
            Z = (S - d[:, None, :] * E) / (d[:, None, :] - d[:, :, None])
        Z[:, idx, idx] = -0.5 * E[:, idx, idx]
        zo = Z.copy(); zo[:, idx, idx] = 0
        zm = np.abs(zo).max((1, 2))
        return Z, np.where(np.isfinite(zm), zm, np.inf)

    VH = V.conj().transpose(0, 2, 1)
    S, Gm = VH @ C @ V, VH @ V
    Z1, z1 = step(S, Gm)
    Z1 = np.where(np.isfinite(Z1), Z1, 0)
    T = I + Z1
    TH = T.conj().transpose(0, 2, 1)
    V2 = V @ T
    S2, G2 = TH @ S @ T, V2.conj().transpose(0, 2, 1) @ V2
    Z2, z2 = step(S2, G2)
    return z1, z2


if __name__ == "__main__":
    C = make_C(2048, kind="bench")
    V2s, _ = two_sided(C)
    z1, z2 = oa_two_steps(C, V2s)
    print("two-step refinement on the bench model:")
    for lim in (1e-3, 3e-3, 1e-2, 3e-2, 1e-1):
        first = z1 <= 3e-5
        second = (~first) & (z1 <= lim) & (z2 <= 3e-5)
        wasted = (~first) & (z1 <= lim) & ~(z2 <= 3e-5)
        print(f"  second step tried for |Z1| <= {lim:g}: pass first {first.mean():.3f}, rescued {second.mean():.3f}, tried in vain {wasted.mean():.4f}, straight to sweeps {((~first) & (z1 > lim)).mean():.4f}")
