"""cfg1-shaped subband stream (8 x 8, bundled rirs): per-bin filter error against the oracle vs conditioning and the
eigenvalue gap at the truncation rank, for the default Jacobi stop threshold and a tight one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ap_vast_unofficial_amd.apvast import apvast
from oracle.subband_stream import SubbandStreamOracle

g = np.load(os.path.join(ROOT, "tests/golden/rirs_cfg1.npz"))
rirA, rirB = g["rirA"], g["rirB"]
for tol2 in (0.0, 1e-14, 1e-18):
    ap = apvast(256, rirA, rirB, 16, 16, 0, 0, 8, 1.0, 1024, hop_size=128, run_B=False, perceptual=False, seed=0, sweep_tol2=tol2)
    rs = np.random.RandomState(0)
    init_r = np.stack([1e-3 * rs.randn(256, 8, 8) for _ in range(4)]); init_t = np.stack([1e-3 * rs.randn(256, 8) for _ in range(2)])
    orc = SubbandStreamOracle(256, rirA, rirB, 16, 0, 0, list(range(1, 9)), 1.0, hop_size=128, run_B=False, init_response=init_r, init_target_response=init_t)
    x = np.random.default_rng(99).standard_normal((2, 4 * 128))
    for h in range(4):
        ap.process_input_buffers(x[0, h*128:(h+1)*128], x[1, h*128:(h+1)*128]); orc.process(x[0, h*128:(h+1)*128], x[1, h*128:(h+1)*128])
    XD = orc.spectra[1].transpose(0, 2, 1)
    RD = np.einsum("kmi,kmj->kij", XD.conj(), XD) + 1e-7 * np.eye(8)
    kap = np.linalg.cond(RD)
    lam, lr = ap.lambda_A, orc.lam[0]
    w, wr = ap.w_A, orc.w[0].transpose(1, 0, 2)
    werr = np.linalg.norm(w - wr, axis=-1) / np.linalg.norm(wr, axis=-1)          # (V, K)
    gap = (lr[:, :-1] - lr[:, 1:]) / lr[:, :1]                                   # relative gap below rank V (V = 1..7)
    lerr = np.abs(lam - lr).max(axis=1) / lr[:, 0]
    print(f"tol2={tol2:g}: lam err max {lerr.max():.1e}; w err max over ranks/bins {werr.max():.1e}, full rank (V=8) max {werr[7].max():.1e}")
    worst = np.argsort(werr[:7].max(axis=0))[-5:]
    for k in worst:
        v = int(np.argmax(werr[:7, k]))
        print(f"   bin {k}: werr {werr[v, k]:.1e} at rank {v+1}, gap below that rank {gap[k, v]:.1e}, kappa(RD) {kap[k]:.1e}, werr*gap {werr[v,k]*gap[k,v]:.1e}")
    ap.close()
