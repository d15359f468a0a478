import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
import bench
K=32*1024
XB,XD,d=bench.synth(K,1234)
for dt in ("f64","f32"):
    res={}
    for ms,stop in ((1,1),(1,2),(1,3),(1,0),(2,0),(0,0)):
        eng=Engine(K,16,32,ranks=(8,),compute_dtype=dt,out_c128=False,max_sweeps=ms,debug_stop=stop)
        dXB,dXD,dd=eng.to_device(XB),eng.to_device(XD),eng.to_device(d)
        dw=eng.alloc(K*16*8); ds=eng.alloc(K*4)
        for _ in range(2): eng.update_dev(dXB,dXD,dd,dw,None,ds)
        eng.sync(); eng.timer_start()
        for _ in range(10): eng.update_dev(dXB,dXD,dd,dw,None,ds)
        ms_t=eng.timer_stop()/10
        st=ds.download((K,),np.int32)
        res[ms]=ms_t
        print(dt,"stop",stop,"max_sweeps",ms,"ms",round(ms_t,4),"status2 frac",float((st==2).mean()))
        eng.close()
