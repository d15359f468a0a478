"""CPU oracle: the per-bin ("subband") AP-VAST update.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference repository has no per-bin code; the
north-star path is the composition, per frequency bin k, of stages the
reference does have:

  X[k]  = control-point spectra  (apvast.py:244-262; X[k] = spectra[k].T, M x L)
  R[k]  = X[k]^H X[k], r[k] = X_B[k]^H d[k]   (complex twin of Y Y^T, Y d at
                                               apvast.py:339-340, 347)
  U,lam = jdiag(R_B[k], R_D[k])               (apvast.py:20-36, 380)
  w[k]  = sum_{i<V} (u_i^H r)/(lam_i+mu) u_i  (apvast.py:406-414)

All arithmetic here is complex128 whatever the input dtype.
"""
import numpy as np

from . import gevd


def correlate(XB, XD, d):
    """R_B, R_D (K,L,L) and r (K,L) from XB, XD (K,M,L) and d (K,M)."""
    XB = np.asarray(XB, dtype=np.complex128)
    XD = np.asarray(XD, dtype=np.complex128)
    d = np.asarray(d, dtype=np.complex128)
    RB = np.einsum("kmi,kmj->kij", XB.conj(), XB)
    RD = np.einsum("kmi,kmj->kij", XD.conj(), XD)
    r = np.einsum("kmi,km->ki", XB.conj(), d)
    return RB, RD, r


def gevd_vast(RB, RD, r, mu, ranks, reg_mode=gevd.REG_MODE_ABS, reg=None,
              reg_bright=0.0):
    """Per-bin jdiag + filter.  Returns w (K,nV,L), lam (K,L), status (K,).

    ``status[k] != 0`` marks a bin whose loaded R_D was not positive definite
    (numpy.linalg.LinAlgError at apvast.py:24); its w and lam are zero.
    ``reg_bright`` is the optional relative bright loading of the MATLAB
    dialect (apVast.m:552-569); 0 in the Python dialect.
    """
    K, L, _ = RB.shape
    w = np.zeros((K, len(ranks), L), dtype=np.complex128)
    lam = np.zeros((K, L))
    status = np.zeros(K, dtype=np.int32)
    for k in range(K):
        A = RB[k]
        if reg_bright:
            A = A + reg_bright * np.linalg.norm(A, ord=2) * np.eye(L)
        try:
            U, lk = gevd.jdiag(A, RD[k], reg_mode, reg)
        except np.linalg.LinAlgError:
            status[k] = 1
            continue
        lam[k] = lk
        w[k] = gevd.vast_filter(U, lk, r[k], mu, ranks)
    return w, lam, status


def update(XB, XD, d, mu, ranks, reg_mode=gevd.REG_MODE_ABS, reg=None):
    """One block of subband filter updates (the unit of BASELINE.json's metric)."""
    RB, RD, r = correlate(XB, XD, d)
    return gevd_vast(RB, RD, r, mu, ranks, reg_mode, reg)


def update_vectorised(XB, XD, d, mu, ranks, reg=gevd.REG_ABS):
    """Same result through batched LAPACK calls (cholesky -> inverse -> eigh).

    The "strongest honest NumPy baseline" of BASELINE.md section 3, step B2.
    """
    RB, RD, r = correlate(XB, XD, d)
    K, L, _ = RB.shape
    Bc = np.linalg.cholesky(RD + reg * np.eye(L))
    Li = np.linalg.inv(Bc)
    C = Li @ RB @ Li.conj().transpose(0, 2, 1)
    C = 0.5 * (C + C.conj().transpose(0, 2, 1))
    lam, Q = np.linalg.eigh(C)
    lam = lam[:, ::-1]
    Q = Q[:, :, ::-1]
    U = Li.conj().transpose(0, 2, 1) @ Q
    coef = np.einsum("kli,kl->ki", U.conj(), r) / (lam + mu)
    w = np.stack([np.einsum("kli,ki->kl", U[:, :, :V], coef[:, :V]) for V in ranks], axis=1)
    return w, lam


# --------------------------------------------------------------------------
# STFT stages (apvast.py:94, 197-225, 244-293, 428-504)
# --------------------------------------------------------------------------

def sine_window(N):
    """apvast.py:94."""
    return np.sin(np.pi / N * np.arange(N))


def analysis(buf, window):
    """rfft(window * buf) along axis 0 (apvast.py:202-203, 246-255, 430-431)."""
    shape = (-1,) + (1,) * (buf.ndim - 1)
    return np.fft.rfft(window.reshape(shape) * buf, axis=0)


def synthesis_ola(spec, window, overlap, hop):
    """overlap <- shift(overlap, hop) + window * irfft(spec)  (apvast.py:265-293, 457-465).

    Returns the new overlap buffer; its first ``hop`` rows are the finished
    samples (apvast.py:299, 500).
    """
    N = window.size
    shape = (-1,) + (1,) * (spec.ndim - 1)
    new = window.reshape(shape) * np.fft.irfft(spec, N, axis=0)
    out = np.zeros_like(overlap)
    out[: N - hop] = overlap[hop:]
    return out + new
