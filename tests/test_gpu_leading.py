"""GPU parity of the leading-eigenpair solver (csrc/kernels_gevd_lead.hip) behind apv_jdiag_leading and the broadband hop:
what apvast.py:406-414 consumes of jdiag's result (apvast.py:20-36, called at 380, 382)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gevd  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def Engine():
    from ap_vast_unofficial_amd import Engine
    return Engine


def _triu_to_full(t, n):
    R = np.zeros((n, n))
    R[np.triu_indices(n)] = t
    return R + np.triu(R, 1).T


def _pencil(lam, seed):
    """(A, B) with A x_i = lam_i (B + 1e-7 I) x_i for prescribed lam: B + 1e-7 I = X^-T X^-1, A = X^-T diag(lam) X^-1."""
    n = lam.size
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Xi = Q * np.exp(rng.uniform(-1.0, 1.0, n))[None, :]        # X^-1 = (Q D)^T: moderately conditioned
    Bl = Xi @ Xi.T
    A = Xi @ np.diag(lam) @ Xi.T
    return 0.5 * (A + A.T), 0.5 * (Bl + Bl.T) - 1e-7 * np.eye(n)


def _check_pairs(A, B, U, lam, lam_ref, tol_lam=1e-9):
    n, V = U.shape
    Bl = B + 1e-7 * np.eye(n)
    assert np.abs(lam / lam_ref[:V] - 1).max() < tol_lam
    assert np.abs(U.T @ Bl @ U - np.eye(V)).max() < 1e-10
    res = np.linalg.norm(A @ U - (Bl @ U) * lam, axis=0)
    assert res.max() < 1e-9 * np.linalg.norm(A, 2) * np.linalg.norm(U, axis=0).max()


def test_leading_on_the_reference_pair(Engine, golden):
    """G1: (R_A_to_A, R_A_to_B) of cfg1's last hop (n = 256, V = 8): eigenvalues 1e-9, every rank's filter 1e-8 of the
    reference's (apvast.py:406-414), and the solver that ran is the subspace iteration."""
    g = golden("g1_broadband_cfg1")
    n, V = 256, 8
    A, B = _triu_to_full(g["R_AA_triu"], n), _triu_to_full(g["R_AB_triu"], n)
    eng = Engine(1, 4, 4)
    U, lam, info = eng.jdiag_leading(A[None], B[None], V)
    eng.close()
    assert info[0] == 0
    U, lam = U[0], lam[0]
    lam_ref = g["lam"][-1, 0]
    _check_pairs(A, B, U, lam, lam_ref)
    r = g["r"][-1, 0]
    coef = (U.T @ r) / (lam + 1.0)
    for i in range(V):
        w = U[:, : i + 1] @ coef[: i + 1]
        e = g["w"][-1, 0, i]
        assert np.linalg.norm(w - e) <= 1e-8 * np.linalg.norm(e), i


@pytest.mark.parametrize("n,rank,batch", [(96, 8, 3), (257, 20, 2), (800, 50, 1), (800, 40, 2)])
def test_leading_vs_oracle(Engine, n, rank, batch):
    """Block widths 32, 48, 64, ghost rows (n not a multiple of 32), a batch: against the oracle's jdiag."""
    rng = np.random.default_rng(n + rank)
    Y = rng.standard_normal((batch, 2 * n, n)) * np.linspace(1.0, 3.0, n)[None, None, :]
    Z = rng.standard_normal((batch, 3 * n, n))
    A = np.einsum("kmi,kmj->kij", Y, Y)
    B = np.einsum("kmi,kmj->kij", Z, Z)
    eng = Engine(1, 4, 4)
    U, lam, info = eng.jdiag_leading(A, B, rank)
    eng.close()
    for k in range(batch):
        Uo, lam_ref = gevd.jdiag(A[k], B[k])
        _check_pairs(A[k], B[k], U[k], lam[k], lam_ref)
        # the leading subspace itself (individual vectors carry a sign)
        Bl = B[k] + 1e-7 * np.eye(n)
        P, Po = U[k] @ U[k].T @ Bl, Uo[:, :rank] @ Uo[:, :rank].T @ Bl
        assert np.abs(P - Po).max() < 1e-8 * np.abs(Po).max()


def test_leading_cluster_straddles_the_cut(Engine):
    """lambda_V = lambda_{V+1} to 1e-13: the subspace below the cut has no gap, so the V-th vector is ANY vector of the
    cluster's eigenspace (the reference's LAPACK picks one by rounding).  What is defined must still be right: the
    eigenvalues, the invariants, the residual of every returned pair, and the span of the first V - 1 vectors."""
    n, V = 192, 12
    lam = np.concatenate([np.linspace(40.0, 12.0, V - 1), [10.0, 10.0 * (1 - 1e-13)], np.linspace(9.0, 0.05, n - V - 1)])
    A, B = _pencil(lam, 5)
    eng = Engine(1, 4, 4)
    U, lv, info = eng.jdiag_leading(A[None], B[None], V)
    eng.close()
    U, lv = U[0], lv[0]
    _check_pairs(A, B, U, lv, lam)
    Bl = B + 1e-7 * np.eye(n)
    Uo, lo = gevd.jdiag(A, B)
    assert np.abs(lo[:V] / lam[:V] - 1).max() < 1e-9
    P, Po = U[:, : V - 1] @ U[:, : V - 1].T @ Bl, Uo[:, : V - 1] @ Uo[:, : V - 1].T @ Bl
    assert np.abs(P - Po).max() < 1e-8 * np.abs(Po).max()
    # the V-th vector lies in the cluster's two-dimensional eigenspace
    E = Uo[:, V - 1: V + 1]
    u = U[:, V - 1]
    assert np.linalg.norm(u - E @ (E.T @ Bl @ u)) < 1e-7 * np.linalg.norm(u)


def test_leading_falls_back_on_a_flat_spectrum(Engine):
    """Eigenvalues within 0.1 % of one another: no polynomial of affordable degree separates the block from the rest, the
    iteration gives up after its pass budget and the complete block-Jacobi solve answers (info = 1), to the same bounds."""
    n, V = 160, 8
    lam = 1.0 + 1e-3 * np.linspace(1.0, 0.0, n)
    A, B = _pencil(lam, 9)
    eng = Engine(1, 4, 4)
    U, lv, info = eng.jdiag_leading(A[None], B[None], V)
    assert info[0] == 1
    _check_pairs(A, B, U[0], lv[0], lam)
    # a rank that leaves fewer than eight guard vectors in a block of 64 goes to the complete solve as well
    lam2 = np.linspace(30.0, 0.1, 240)
    A2, B2 = _pencil(lam2, 10)
    U2, lv2, info2 = eng.jdiag_leading(A2[None], B2[None], 60)
    eng.close()
    assert info2[0] == 1
    _check_pairs(A2, B2, U2[0], lv2[0], lam2)
    # a dark matrix that is not positive definite raises as the reference's cholesky does (apvast.py:21-24): on this path the
    # factorisation's flag comes back with the subspace iteration's first pass
    eng3 = Engine(1, 4, 4)
    with pytest.raises(np.linalg.LinAlgError):
        eng3.jdiag_leading(np.eye(128)[None], -np.eye(128)[None], 8)
    eng3.close()


def test_hop_attributes_complete_themselves_when_read(golden):
    """The per-hop call computes the leading block only; lambda_* and U_* in full (apvast.py:380-387) are worked out from the
    hop's matrices when they are read, and agree with the reference's on every one of the n eigenvalues."""
    from ap_vast_unofficial_amd.apvast import apvast
    g, rirs = golden("g1_broadband_cfg1"), golden("rirs_cfg1")
    x, H = g["x"], 128
    ap = apvast(256, rirs["rirA"], rirs["rirB"], 32, 16, 0, 0, 8, 1.0, 512, hop_size=H, perceptual=False, mode="broadband", seed=0)
    ap.set_state({"response": g["init_response"], "target_response": g["init_target_response"]})
    for h in range(8):
        ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    for z, name in enumerate(("lambda_A", "lambda_B")):
        lam = np.asarray(getattr(ap, name)).ravel()
        ref = g["lam"][-1, z]
        assert lam.shape == ref.shape
        assert np.abs(lam - ref).max() < 1e-9 * ref[0]
        assert np.abs(lam[:8] / ref[:8] - 1).max() < 1e-9
    U = np.asarray(ap.U_A)
    A, B = _triu_to_full(g["R_AA_triu"], 256), _triu_to_full(g["R_AB_triu"], 256)
    assert np.abs(U.T @ (B + 1e-7 * np.eye(256)) @ U - np.eye(256)).max() < 1e-9
    assert np.abs(U.T @ A @ U - np.diag(np.asarray(ap.lambda_A).ravel())).max() < 1e-9 * g["lam"][-1, 0, 0]
