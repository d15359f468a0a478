#!/usr/bin/env python3
"""BASELINE config 5, correlation stage alone: 64 loudspeakers x 128 control points x 2048 bins.
fp32 accumulation (exact-product f32 MFMA) vs bf16 inputs (bf16 MFMA, f32 accumulation) vs the f64 VALU kernel:
time, algorithmic GB/s against the 8 TB/s HBM roof, error against the float64 oracle."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ap_vast_unofficial_amd import Engine
from oracle import subband


def main():
    L, M, K = 64, 128, 2048
    rng = np.random.default_rng(1234)
    def cn(*s):
        o = np.empty(s, np.complex64)
        o.real = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(.5))
        o.imag = rng.standard_normal(s, dtype=np.float32) * np.float32(np.sqrt(.5))
        return o
    XB, XD, d = cn(K, M, L), cn(K, M, L), cn(K, M)
    ks = slice(0, 32)
    RB0, RD0, r0 = subband.correlate(XB[ks], XD[ks], d[ks])
    out = {"workload": "cfg5 correlation 64x128x2048", "hbm_peak_GBps": 8000.0}
    lib_bytes = {"f32": 2 * K * M * L * 8 + K * M * 8 + 2 * K * L * L * 8 + K * L * 8,
                 "bf16": 2 * K * M * L * 4 + K * M * 4 + 2 * K * L * L * 8 + K * L * 8,
                 "f64": 2 * K * M * L * 8 + K * M * 8 + 2 * K * L * L * 16 + K * L * 16}
    for mode in ("f32", "bf16", "f64"):
        eng = Engine(K, L, M, compute_dtype="f64" if mode == "f64" else "f32")
        src = [eng.to_device(a) for a in (XB, XD, d)]
        cs = 16 if mode == "f64" else 8
        dRB, dRD, dr = eng.alloc(K * L * L * cs), eng.alloc(K * L * L * cs), eng.alloc(K * L * cs)
        if mode == "bf16":
            bf = [eng.alloc(n * 4) for n in (K * M * L, K * M * L, K * M)]
            for s_, b_, n in zip(src, bf, (K * M * L, K * M * L, K * M)):
                eng._chk(eng.lib.apv_to_bf16_dev(eng.h, n, s_.ptr, b_.ptr))
            run = lambda: eng._chk(eng.lib.apv_corr_bf16_dev(eng.h, bf[0].ptr, bf[1].ptr, bf[2].ptr, dRB.ptr, dRD.ptr, dr.ptr))
        else:
            run = lambda: eng._chk(eng.lib.apv_corr_dev(eng.h, src[0].ptr, src[1].ptr, src[2].ptr, dRB.ptr, dRD.ptr, dr.ptr))
        for _ in range(3):
            run()
        eng.sync()
        eng.timer_start()
        for _ in range(20):
            run()
        ms = eng.timer_stop() / 20
        dt = np.complex128 if mode == "f64" else np.complex64
        RB = dRB.download((K, L, L), dt)[ks]
        RD = dRD.download((K, L, L), dt)[ks]
        r = dr.download((K, L), dt)[ks]
        fro = lambda a, b: float((np.linalg.norm((a - b).reshape(a.shape[0], -1), axis=1) / np.linalg.norm(b.reshape(b.shape[0], -1), axis=1)).max())
        out[mode] = {"ms": ms, "algorithmic_GBps": lib_bytes[mode] / ms / 1e6, "frac_of_hbm_peak": lib_bytes[mode] / ms / 1e6 / 8000.0,
                     "bins_per_s": K / ms * 1e3, "R_B_rel_fro_err": fro(RB, RB0), "R_D_rel_fro_err": fro(RD, RD0), "r_rel_err": fro(r, r0)}
        eng.close()
    print(json.dumps(out))

main()
