import numpy as np, sys
sys.path.insert(0,'.')
from ap_vast_unofficial_amd import _capi
from oracle import gevd
for n in (96, 200, 512):
    rng = np.random.default_rng(n)
    Y = rng.standard_normal((1, 2 * n, n)) + 1j * rng.standard_normal((1, 2 * n, n))
    Z = rng.standard_normal((1, 2 * n, n)) + 1j * rng.standard_normal((1, 2 * n, n))
    A = np.einsum("kmi,kmj->kij", Y.conj(), Y); B = np.einsum("kmi,kmj->kij", Z.conj(), Z)
    eng = _capi.Engine(1, 4, 4)
    import time; t0=time.perf_counter()
    U, lam = eng.jdiag_large_complex(A, B); dt=time.perf_counter()-t0
    eng.close()
    _, lr = gevd.jdiag(A[0], B[0])
    G = U[0].conj().T @ (B[0] + 1e-7*np.eye(n)) @ U[0]
    D = U[0].conj().T @ A[0] @ U[0]
    print(n, "lam", np.abs(lam[0]/lr-1).max(), "G", np.abs(G-np.eye(n)).max(), "D", np.abs(D-np.diag(lam[0])).max()/lam[0,0], f"{dt*1e3:.1f} ms")
