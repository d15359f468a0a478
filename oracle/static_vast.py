"""CPU oracle: Matlab/ControlMethods/vast.m and predictPressure.m restated in NumPy.

TEST INFRASTRUCTURE ONLY.  Parity UNPINNED: MATLAB/Octave are absent, so this follows the .m files by reading
(vast.m:46-91, predictPressure.m:12-16); the reference ships no fixture for either.
"""
import numpy as np
import scipy.linalg as sla
import scipy.signal


def predict_pressure(x, rirs):
    """predictPressure.m:12-16: sum over sources of filter(rir(:, s, m), 1, x(:, s))."""
    T, L = x.shape
    P, _, M = rirs.shape
    out = np.zeros((T, M))
    for m in range(M):
        for s in range(L):
            out[:, m] += scipy.signal.lfilter(rirs[:, s, m], 1.0, x[:, s])
    return out


def vast_statistics(gB, gD, J, delay, ref):
    """vast.m:46-77 (ref 0-based).  The impulse drive makes the regressor at step t the RIR delayed by the tap
    index: y[s*J + j](t) = g_s[t - 1 - j]; steps t = 0..999."""
    Nb, P, L = gB.shape
    n = J * L
    N = 1000

    def regressors(g):                       # (Nm, N, n)
        Nm = g.shape[0]
        Y = np.zeros((Nm, N, n))
        for t in range(N):
            for j in range(J):
                q = t - 1 - j
                if 0 <= q < P:
                    Y[:, t, np.arange(L) * J + j] = g[:, q, :]
        return Y

    YB, YD = regressors(gB), regressors(gD)
    dref = np.zeros((Nb, P))
    dref[:, delay:] = gB[:, : P - delay, ref]
    d = np.zeros((Nb, N))
    for t in range(1, N):
        if t - 1 < P:
            d[:, t] = dref[:, t - 1]
    RB = np.einsum("mti,mtj->ij", YB, YB)
    RD = np.einsum("mti,mtj->ij", YD, YD)
    rB = np.einsum("mti,mt->i", YB, d)
    f = 1.0 / (Nb * (P - J))
    return RB * f, RD * f, rB * f


def vast(gB, gD, J, delay, ref, V, mu):
    """vast.m:85-91: jdiag(RB, RD, 'vector', true) = eig(A, B, 'chol'), descending; no loading."""
    RB, RD, rB = vast_statistics(gB, gD, J, delay, ref)
    lam, U = sla.eigh(RB, RD)
    lam, U = lam[::-1], U[:, ::-1]
    w = np.zeros(J * gB.shape[2])
    for i in range(V):
        w += (U[:, i] @ rB) / (lam[i] + mu) * U[:, i]
    return w, (RB, RD, rB)
