#!/usr/bin/env python3
"""Where does a wave of the order-16 kernel spend its life?  In-kernel stamps (s_memtime of lane 0 at the stage boundaries,
diagnostic instantiation of gevd16m_kernel) on the bench workload: 32 768 bins, 16 x 32, float64.  Prints the median and the
quartiles of every stage's duration in shader cycles and in microseconds (at the clock the launch sustained), the wave's whole
life, and how many waves were alive at once (from the stamps of start and end)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
K, L, M = int(os.environ.get("K", 32 * 1024)), 16, 32
rng = np.random.default_rng(1234)
def cn(*s): return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)
XB, XD, d = cn(K, M, L), cn(K, M, L), cn(K, M)
eng = Engine(K, L, M, ranks=(8,), compute_dtype="f64", out_c128=False)
dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
dw, ds = eng.alloc(K * 16 * 8), eng.alloc(K * 4)
dst = eng.alloc(K * 16 * 8)
for _ in range(30): eng.update_dev(dXB, dXD, dd, dw, None, ds)
eng.sync(); eng.timer_start()
for _ in range(50): eng.update_dev(dXB, dXD, dd, dw, None, ds)
ms_plain = eng.timer_stop() / 50
eng.debug_set_stamps(dst)
for _ in range(30): eng.update_dev(dXB, dXD, dd, dw, None, ds)
eng.sync(); eng.timer_start()
for _ in range(50): eng.update_dev(dXB, dXD, dd, dw, None, ds)
ms = eng.timer_stop() / 50
st16 = dst.download((K, 16), np.uint64).astype(np.int64)
st = st16[:, :8]
eng.debug_set_stamps(None)
names = ["correlate (2 slabs: loads, 48 MFMA)", "f64 Cholesky + inverse", "whitening (24 MFMA)", "float Cholesky of C", "one-sided sweeps",
         "refinement (48+ MFMA)", "sort, back-transform, filter, stores"]
dur = np.diff(st, axis=1)
life = st[:, 7] - st[:, 0]
ghz = float(os.environ.get("GHZ", "2.4"))     # s_memtime counters of different XCDs are not aligned: no span across waves; nominal clock
print(f"# stage stamps of gevd16m_kernel_f64<fused> (diagnostic instantiation), K = {K}\n")
print(f"launch: {ms:.4f} ms with stamps, {ms_plain:.4f} ms product instantiation; {K / 1024:.0f} waves per SIMD over the launch; microseconds at {ghz} GHz\n")
print("| stage | median cycles | quartiles | median us | share of the wave's life |\n|---|---|---|---|---|")
for i, n in enumerate(names):
    q = np.percentile(dur[:, i], [25, 50, 75])
    print(f"| {n} | {q[1]:.0f} | {q[0]:.0f} - {q[2]:.0f} | {q[1] / ghz / 1e3:.2f} | {100 * dur[:, i].sum() / life.sum():.1f} % |")
q = np.percentile(life, [25, 50, 75])
print(f"| whole life | {q[1]:.0f} | {q[0]:.0f} - {q[2]:.0f} | {q[1] / ghz / 1e3:.2f} | 100 % |")
fine = [("start -> loads of X_B, d landed", st16[:, 8] - st16[:, 0]), ("24 MFMA of X_B + r", st16[:, 9] - st16[:, 8]),
        ("Im R = P - P^T through LDS", st16[:, 10] - st16[:, 9]), ("loads of X_D landed", st16[:, 11] - st16[:, 10]),
        ("24 MFMA of X_D", st16[:, 12] - st16[:, 11]), ("Im R through LDS, sync", st16[:, 1] - st16[:, 12])]
print("\ninside the correlate stage (the diagnostic build waits for the loads before the first MFMA):\n\n| step | median cycles | quartiles |\n|---|---|---|")
for n, v in fine:
    q = np.percentile(v, [25, 50, 75])
    print(f"| {n} | {q[1]:.0f} | {q[0]:.0f} - {q[2]:.0f} |")
print(f"\nsum of the waves' lives / (1024 SIMDs x launch time) = {life.sum() / ghz / 1e3 / (1024 * ms * 1e3):.2f} waves per SIMD alive on average")
eng.close()
