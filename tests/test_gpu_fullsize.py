"""Parity at BASELINE.json's full sizes through size-independent properties (plus the vectorised oracle where it
finishes in seconds):
  cfg2  1024 bins x 32 blocks, 16 x 32      cfg4 shard  512 bins, 16 x 32      cfg5  2048 bins, 64 x 128
Properties: KA-4 pressure-matching limit (V = L  =>  w = (R_B + mu (R_D + reg I))^-1 r), rank-1 limit, eigenvalue
order, phase invariance of w, linearity in d, idempotence of a repeated launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import subband  # noqa: E402


def cn(rng, *s):
    out = np.empty(s, dtype=np.complex64)
    out.real = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(0.5))
    out.imag = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(0.5))
    return out


def pressure_matching(XB, XD, d, mu, reg):
    RB, RD, r = subband.correlate(XB, XD, d)
    L = RB.shape[-1]
    return np.linalg.solve(RB + mu * (RD + reg * np.eye(L)), r[..., None])[..., 0]


@pytest.mark.parametrize("K,L,M,dtype", [(32 * 1024, 16, 32, "f64"), (32 * 1024, 16, 32, "f32"), (512, 16, 32, "f64"),
                                         (2048, 64, 128, "f32"), (2048, 64, 128, "f64")])
def test_full_size_properties(K, L, M, dtype):
    from ap_vast_unofficial_amd import Engine
    rng = np.random.default_rng(K + L)
    XB, XD, d = cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M)
    mu, reg = 1.0, 1e-7
    eng = Engine(K, L, M, ranks=(1, L // 2, L), mu=mu, compute_dtype=dtype, reg_dark=reg)
    w, lam, status = eng.update(XB, XD, d)
    w2, lam2, _ = eng.update(XB, XD, d)
    assert not status.any()
    assert np.array_equal(w, w2) and np.array_equal(lam, lam2)                 # idempotent / deterministic
    assert (np.diff(lam, axis=1) <= 0).all() and (lam[:, -1] > 0).all()        # descending, positive
    tol = 1e-7 if dtype == "f64" else (2e-3 if L == 64 else 2e-4)
    # KA-4 on every bin
    pm = pressure_matching(XB, XD, d, mu, reg)
    err = np.linalg.norm(w[:, 2] - pm, axis=1) / np.linalg.norm(pm, axis=1)
    assert err.max() < tol, err.max()
    # vectorised oracle on every bin (all three ranks)
    w_ref, lam_ref = subband.update_vectorised(XB, XD, d, mu, [1, L // 2, L], reg=reg)
    lt = 1e-9 if dtype == "f64" else 1e-5
    assert (np.abs(lam - lam_ref) / lam_ref[:, :1]).max() < lt * (10 if (L == 64 and dtype == "f32") else 1)
    errw = np.linalg.norm(w - w_ref, axis=-1) / np.linalg.norm(w_ref, axis=-1)
    if dtype == "f64":
        assert errw.max() < tol, errw.max()
    else:
        # float32, partial ranks: w_V = sum_{i<V} (u_i^H r) / (lam_i + mu) u_i loses relative accuracy in the bins where r
        # is nearly orthogonal to the leading eigenvectors (cancellation in u_i^H r) or where lam_V ~ lam_V+1; among
        # 32 768 random bins the worst case lands at 2-3e-4 whichever kernel runs (99.9th percentile 2e-5, mean 1e-6).
        # The full-rank filter (no cut, KA-4 above) keeps the plain tolerance.
        assert errw[:, 2].max() < tol, errw[:, 2].max()
        assert np.percentile(errw, 99.9) < tol / 4 and errw.max() < 5 * tol, (np.percentile(errw, 99.9), errw.max())
    # linearity in d and phase invariance: d -> c d scales w by c (c = -2j is exact in binary floating point, so
    # the inputs of the two runs are exact multiples of each other)
    c = np.complex64(-2j)
    wc, _, _ = eng.update(XB, XD, c * d)
    eng.close()
    errc = np.linalg.norm(wc - c * w, axis=-1) / np.linalg.norm(w, axis=-1)
    assert errc.max() < (1e-12 if dtype == "f64" else tol), errc.max()
