import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
from oracle import subband
L,M,K=64,128,256
rng=np.random.default_rng(1234)
def cn(*s): return ((rng.standard_normal(s)+1j*rng.standard_normal(s))*np.sqrt(.5)).astype(np.complex64)
XB,XD,d=cn(K,M,L),cn(K,M,L),cn(K,M)
_,_,r0=subband.correlate(XB,XD,d)
eng=Engine(K,L,M,compute_dtype="f32")
for t in range(3):
    RB,RD,r=eng.corr_bf16(XB,XD,d)
    rel=np.linalg.norm(r-r0,axis=1)/np.linalg.norm(r0,axis=1)
    print("run",t,"bad bins",np.where(rel>0.01)[0].tolist())
RBf,RDf,rf=eng.corr(XB,XD,d)
rel=np.linalg.norm(rf-r0,axis=1)/np.linalg.norm(r0,axis=1)
print("f32 mfma r: max rel", rel.max())
