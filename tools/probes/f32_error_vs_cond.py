import numpy as np, sys
sys.path.insert(0,'.')
from ap_vast_unofficial_amd import Engine
from oracle import subband
def w_err(w, ref): return (np.linalg.norm(w-ref,axis=-1)/np.linalg.norm(ref,axis=-1)).max()
for name in ("g3_jdiag_c_16x32","g3_jdiag_c_8x8","g3_jdiag_c_64x128"):
    g=np.load(f"tests/golden/{name}.npz")
    XB,XD,d=g["XB"],g["XD"],g["d"]; K,M,L=XB.shape
    ranks=[int(v) for v in g["ranks"]]
    eng=Engine(K,L,M,ranks=ranks,mu=float(g["mu"]),compute_dtype="f32",reg_dark=float(g["reg"]))
    w,lam,st=eng.update(XB,XD,d); eng.close()
    RD=subband.correlate(XB,XD,d)[1]+float(g["reg"])*np.eye(L)
    cond=np.linalg.cond(RD)
    el=(np.abs(lam-g["lam"])/g["lam"][:,:1]).max(axis=1)
    ew=(np.linalg.norm(w-g["w"],axis=-1)/np.linalg.norm(g["w"],axis=-1)).max(axis=1)
    eps=np.finfo(np.float32).eps
    print(name, "cond max %.3g"%cond.max(), "lam err max %.3g"%el.max(), "w err max %.3g"%ew.max(), "max lam/(eps*cond) %.3g"%(el/(eps*cond)).max(), "max w/(eps*cond) %.3g"%(ew/(eps*cond)).max())
