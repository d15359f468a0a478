"""Is a block-Jacobi round of the large solver bound by its pair-solve chain or by the chip?  Times apv_jdiag_large at
n = 800 (and 256) for batch 1, 2, 4, 8 (host copies included: 2 x n^2 x 8 B x batch each way, ~1 ms per matrix at n = 800)."""
import json, sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ap_vast_unofficial_amd._capi import Engine

def main():
    rng = np.random.default_rng(3)
    eng = Engine(1, 4, 4)
    for n in (256, 800):
        for batch in (1, 2, 4, 8, 16):
            Y = rng.standard_normal((2, batch, n, 2 * n))
            A = Y[0] @ Y[0].transpose(0, 2, 1)
            B = Y[1] @ Y[1].transpose(0, 2, 1)
            eng.jdiag_large(A, B)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                eng.jdiag_large(A, B)
            ms = (time.perf_counter() - t0) / reps * 1e3
            print(json.dumps({"n": n, "batch": batch, "ms_per_call": round(ms, 3), "ms_per_matrix": round(ms / batch, 3)}), flush=True)

main()
