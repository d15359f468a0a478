#!/usr/bin/env python3
"""Where does a workgroup of the order-64 two-bin kernel (gevd64x2_kernel) spend its life?  s_memtime of thread 0 at the phase
boundaries (apv_debug_set_stamps) on BASELINE config 5: 2048 bins, 64 x 128, float64.  Medians over the 1024 workgroups."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine
K, L, M = int(os.environ.get("K", 2048)), 64, 128
rng = np.random.default_rng(1234)
def cn(*s): return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)
XB, XD, d = cn(K, M, L), cn(K, M, L), cn(K, M)
eng = Engine(K, L, M, ranks=(1, 32, 64), compute_dtype="f64", out_c128=False)
dXB, dXD, dd = eng.to_device(XB), eng.to_device(XD), eng.to_device(d)
dw, ds = eng.alloc(K * 3 * L * 8), eng.alloc(K * 4)
dst = eng.alloc(K * 16 * 8)
def run(n):
    eng.update_dev(dXB, dXD, dd, dw, None, ds); eng.sync(); eng.timer_start()
    for _ in range(n): eng.update_dev(dXB, dXD, dd, dw, None, ds)
    return eng.timer_stop() / n
ms_plain = run(4)
eng.debug_set_stamps(dst)
ms = run(4)
st = dst.download((K, 16), np.uint64).astype(np.int64)
eng.debug_set_stamps(None)
ms_plain2 = run(6)
ghz = float(os.environ.get("GHZ", "2.4"))
a, b = st[0::2], st[1::2]                       # rows of the first / second bin of every workgroup
rows = [("front stages, first bin", a[:, 1] - a[:, 0]),
        ("  correlate", a[:, 8] - a[:, 0]), ("  Cholesky", a[:, 9] - a[:, 8]), ("  inverse", a[:, 10] - a[:, 9]),
        ("  whitening, stores to the slot", a[:, 1] - a[:, 10]),
        ("front stages, second bin", a[:, 2] - a[:, 1]),
        ("  correlate", b[:, 8] - a[:, 1]), ("  Cholesky", b[:, 9] - b[:, 8]), ("  inverse", b[:, 10] - b[:, 9]),
        ("float sweeps of both bins, interleaved (incl. loads of C)", a[:, 3] - a[:, 2]),
        ("eigenvector matrices to the slots", a[:, 4] - a[:, 3]),
        ("back stages, first bin", a[:, 5] - a[:, 4]), ("  refinement", a[:, 11] - a[:, 4]), ("  sort, back-transform, filters", a[:, 5] - a[:, 11]),
        ("back stages, second bin", a[:, 6] - a[:, 5]), ("  refinement", b[:, 11] - a[:, 5]),
        ("whole life of the workgroup", a[:, 6] - a[:, 0])]
print(f"# phase stamps of gevd64x2_kernel<fused>, K = {K} (two bins per workgroup)\n")
print(f"launch: {ms:.3f} ms with stamps, {ms_plain:.3f} ms without before them (clocks still ramping), {ms_plain2:.3f} ms without after; microseconds at {ghz} GHz\n")
print("| phase | median cycles | quartiles | median us |\n|---|---|---|---|")
for n, v in rows:
    q = np.percentile(v, [25, 50, 75])
    print(f"| {n} | {q[1]:.0f} | {q[0]:.0f} - {q[2]:.0f} | {q[1] / ghz / 1e3:.1f} |")
print("\ninside the sweeps (sums over the steps of a workgroup, medians over the workgroups):\n\n| what | cycles | us |\n|---|---|---|")
for n, v in (("wave 0 at work: pair solves, every other step", a[:, 12]), ("wave 4 at work: a C tile (two products, mirrored) + a V tile, every other step", a[:, 13]),
             ("wave 15 at work: three V tiles, every other step", a[:, 14]), ("all steps, barriers included", a[:, 15])):
    print(f"| {n} | {np.median(v):.0f} | {np.median(v) / ghz / 1e3:.1f} |")
life = a[:, 6] - a[:, 0]
print(f"\nsum of the workgroups' lives / (256 CUs x launch time) = {life.sum() / ghz / 1e3 / (256 * ms * 1e3):.2f}")
eng.close()
