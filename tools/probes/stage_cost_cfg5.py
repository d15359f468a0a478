import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
L,M,K=64,128,2048
rng=np.random.default_rng(1234)
def cn(*s):
    o=np.empty(s,np.complex64); o.real=rng.standard_normal(s,dtype=np.float32)*np.float32(np.sqrt(.5)); o.imag=rng.standard_normal(s,dtype=np.float32)*np.float32(np.sqrt(.5)); return o
XB,XD,d=cn(K,M,L),cn(K,M,L),cn(K,M)
for dt in ("f32","f64"):
    for ms,stop in ((1,1),(1,2),(1,3),(1,0),(2,0),(0,0)):
        eng=Engine(K,L,M,ranks=(32,),compute_dtype=dt,out_c128=False,max_sweeps=ms,debug_stop=stop)
        dXB,dXD,dd=eng.to_device(XB),eng.to_device(XD),eng.to_device(d)
        dw=eng.alloc(K*L*8); ds=eng.alloc(K*4)
        eng.update_dev(dXB,dXD,dd,dw,None,ds); eng.sync(); eng.timer_start()
        for _ in range(3): eng.update_dev(dXB,dXD,dd,dw,None,ds)
        t=eng.timer_stop()/3
        print(dt,"stop",stop,"max_sweeps",ms,"ms",round(t,3)); eng.close()
