"""CPU oracle: joint diagonalisation (GEVD) and the variable-span filter.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in float64 / complex128 NumPy:
  * ``jdiag``            reference Python/apvast.py:20-36 (spec of record
                         Matlab/ControlMethods/jdiag.m:103-117)
  * ``vast_filter``      reference Python/apvast.py:406-414 (rank-accumulated
                         filter), with the conjugating inner product that the
                         complex (subband) case needs -- SURVEY.md section 3.4,
                         last row.  For real data it is identical to the
                         reference's ``np.inner``.
"""
import numpy as np
import scipy.linalg as sla

# reference Python/apvast.py:7 and :22-27
REG_ABS = 1e-7          # EXPERIMENTAL_REGULARIZATION=True : B + 1e-7 I
REG_REL = 1e-8          # else                             : B + 1e-8 ||B||_2 I

REG_MODE_ABS = 0
REG_MODE_REL = 1


def jdiag(A, B, reg_mode=REG_MODE_ABS, reg=None):
    """Return (U, lam): U^H (B+reg I) U = I, U^H A U = diag(lam), lam descending.

    Follows apvast.py:20-36 step by step: lower Cholesky factor of the loaded
    B (line 24 / 26-27), the two triangular solves that form
    C = Bc^-1 A Bc^-H (lines 28-29), an eigendecomposition of C (line 30: the
    reference calls ``schur``; for the Hermitian C that IS its
    eigendecomposition), the back-substitution X = Bc^-H Q (line 31) and the
    descending sort (lines 32-35).  ``lam`` is returned as a vector (the
    reference wraps it in ``np.diag`` at line 34 and unwraps it at 385/387).
    Raises numpy.linalg.LinAlgError when the loaded B is not positive definite
    (comment at apvast.py:21).
    """
    A = np.asarray(A)
    B = np.asarray(B)
    n = B.shape[0]
    if reg_mode == REG_MODE_ABS:
        load = REG_ABS if reg is None else reg
    else:
        load = (REG_REL if reg is None else reg) * np.linalg.norm(B, ord=2)
    Bc = np.linalg.cholesky(B + load * np.eye(n))
    C0 = sla.solve_triangular(Bc, A, lower=True)
    # right-multiply by Bc^-H:  C = (Bc^-1 (C0)^H)^H
    C = sla.solve_triangular(Bc, C0.conj().T, lower=True).conj().T
    C = 0.5 * (C + C.conj().T)
    lam, Q = np.linalg.eigh(C)
    order = np.argsort(lam)[::-1]
    lam = lam[order]
    Q = Q[:, order]
    X = sla.solve_triangular(Bc.conj().T, Q, lower=False)
    return X, lam


def vast_filter(U, lam, r, mu, ranks):
    """w_V = sum_{i<V} (u_i^H r) / (lam_i + mu) u_i for every V in ``ranks``.

    apvast.py:406-414 accumulates exactly this sum one eigenvector at a time
    (w[i] = w[i-1] + ...).  Returns an array (len(ranks), n).
    """
    r = np.asarray(r).reshape(-1)
    coef = (U.conj().T @ r) / (lam + mu)
    out = np.zeros((len(ranks), U.shape[0]), dtype=np.result_type(U, r))
    for t, V in enumerate(ranks):
        out[t] = U[:, :V] @ coef[:V]
    return out


def jdiag_batched_loop(A, B, reg_mode=REG_MODE_ABS, reg=None):
    """Per-bin loop over ``jdiag`` (the structure the reference would have)."""
    K, n, _ = A.shape
    U = np.empty((K, n, n), dtype=np.result_type(A, B, np.float64))
    lam = np.empty((K, n))
    for k in range(K):
        U[k], lam[k] = jdiag(A[k], B[k], reg_mode, reg)
    return U, lam
