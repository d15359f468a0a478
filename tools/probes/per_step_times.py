import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from ap_vast_unofficial_amd import Engine
import bench
K = 32 * 1024
XB, XD, d = bench.synth(K, 1234)
eng = Engine(K, 16, 32, ranks=(8,), compute_dtype="f64", out_c128=False)
dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
dw, ds = eng.alloc(K * 16 * 8), eng.alloc(K * 4)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for _ in range(8): eng.update_dev(dXB, dXD, dd, dw, None, ds)
eng.sync(); eng.device_sync()
ts = []
for i in range(60):
    eng.timer_start(); eng.update_dev(dXB, dXD, dd, dw, None, ds); ts.append(eng.timer_stop())
print("per-step ms (each its own event pair, back to back):", " ".join("%.3f" % t for t in ts))
eng.sync()
eng.timer_start()
for i in range(20): eng.update_dev(dXB, dXD, dd, dw, None, ds)
a = eng.timer_stop() / 20
eng.timer_start()
for i in range(200): eng.update_dev(dXB, dXD, dd, dw, None, ds)
b = eng.timer_stop() / 200
eng.timer_start()
for i in range(20): eng.update_dev(dXB, dXD, dd, dw, None, ds)
c = eng.timer_stop() / 20
print("20 steps %.4f, then 200 steps %.4f, then 20 steps %.4f" % (a, b, c))
