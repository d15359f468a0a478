import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
from oracle import subband
L,M,K=64,128,256
rng=np.random.default_rng(1234)
def cn(*s): return ((rng.standard_normal(s)+1j*rng.standard_normal(s))*np.sqrt(.5)).astype(np.complex64)
XB,XD,d=cn(K,M,L),cn(K,M,L),cn(K,M)
eng=Engine(K,L,M,compute_dtype="f32")
RB,RD,r=eng.corr_bf16(XB,XD,d)
_,_,r0=subband.correlate(XB,XD,d)
rel=np.linalg.norm(r-r0,axis=1)/np.linalg.norm(r0,axis=1)
print("r rel err: median",np.median(rel),"max",rel.max(),"argmax",rel.argmax())
k=rel.argmax(); e=np.abs(r[k]-r0[k]); print(np.round(e,3))
# rounded-input reference
import struct
def bf(x):
    u=x.view(np.uint32); u=((u+0x7fff+((u>>16)&1))>>16)<<16; return u.astype(np.uint32).view(np.float32)
def cb(a): return bf(a.real.astype(np.float32).copy())+1j*bf(a.imag.astype(np.float32).copy())
_,_,r1=subband.correlate(cb(XB),cb(XD),cb(d))
print("vs rounded-input reference:", (np.linalg.norm(r-r1,axis=1)/np.linalg.norm(r1,axis=1)).max())
bad=np.where(rel>0.01)[0]; print("bad bins", bad[:40], len(bad))
colerr=np.abs(r-r1)
print("bad columns per bad bin:", [np.where(colerr[k]>0.2)[0].tolist() for k in bad[:6]])
RB0,RD0,_=subband.correlate(cb(XB),cb(XD),cb(d))
print("RB err max", np.abs(RB-RB0).max(), "RD", np.abs(RD-RD0).max())
