import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
L,M,K=64,128,256
rng=np.random.default_rng(1234)
def cn(*s): return ((rng.standard_normal(s)+1j*rng.standard_normal(s))*np.sqrt(.5)).astype(np.complex64)
XB,XD,d=cn(K,M,L),cn(K,M,L),cn(K,M)
eng=Engine(K,L,M,compute_dtype="f32")
src=[eng.to_device(a) for a in (XB,XD,d)]
cnt=(K*M*L,K*M*L,K*M)
bf=[eng.alloc(c*4) for c in cnt]
for s_,b_,c_ in zip(src,bf,cnt): eng._chk(eng.lib.apv_to_bf16_dev(eng.h,c_,s_.ptr,b_.ptr))
eng.sync()
dbf=bf[2].download((K*M,),np.uint32); xbf=bf[0].download((K*M*L,),np.uint32)
def tof(u): return (u.astype(np.uint32)<<16).view(np.float32), (u & np.uint32(0xffff0000)).view(np.float32)
dr_,di_=tof(dbf); xr_,xi_=tof(xbf)
dd=(dr_+1j*di_).reshape(K,M).astype(np.complex128); xx=(xr_+1j*xi_).reshape(K,M,L).astype(np.complex128)
r_ref=np.einsum("kmi,km->ki",xx.conj(),dd)
dRB,dRD,dr=eng.alloc(K*L*L*8),eng.alloc(K*L*L*8),eng.alloc(K*L*8)
prev=None
for t in range(4):
    eng._chk(eng.lib.apv_corr_bf16_dev(eng.h,bf[0].ptr,bf[1].ptr,bf[2].ptr,dRB.ptr,dRD.ptr,dr.ptr)); eng.sync()
    r=dr.download((K,L),np.complex64)
    rel=np.linalg.norm(r-r_ref,axis=1)/np.linalg.norm(r_ref,axis=1)
    print("run",t,"bad",np.where(rel>1e-4)[0].tolist(), "max rel", rel.max(), "same as prev", None if prev is None else bool((r==prev).all()))
    prev=r
