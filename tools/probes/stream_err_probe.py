"""Prints the achieved errors of the streaming subband path against the oracle (sets the tolerances of tests/test_gpu_stream.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import test_gpu_stream as T


def report(tag, ap, orc, got, exp, K, L, M, zones):
    e = ap._eng
    out = [tag]
    for p in range(4):
        if (p < 2 and 0 not in zones) or (p >= 2 and 1 not in zones):
            continue
        X = e.get_state(f"spectra{p}", (K, M, L), e.sc_dtype)
        ref = orc.spectra[p].transpose(0, 2, 1)
        out.append(f"spec{p}={np.abs(X - ref).max() / np.abs(ref).max():.1e}")
    for z in zones:
        name = "AB"[z]
        w, wr = getattr(ap, "w_" + name), orc.w[z].transpose(1, 0, 2)
        err = np.linalg.norm(w - wr, axis=-1) / np.linalg.norm(wr, axis=-1)
        lam, lr = getattr(ap, "lambda_" + name), orc.lam[z]
        V = w.shape[0]
        lerr = np.abs(lam[:, :V] - lr[:, :V]).max(axis=1) / lr[:, 0]
        out.append(f"w{name}: med={np.median(err):.1e} max={err.max():.1e} lam={lerr.max():.1e}")
    worst = [0, 0]
    for g, ex in zip(got, exp):
        for q in range(4):
            if ex[q] is None:
                continue
            ref = ex[q] if q < 2 else np.broadcast_to(ex[q], (len(g[q]),) + ex[q].shape)
            arr = np.stack(g[q])
            worst[q >= 2] = max(worst[q >= 2], np.abs(arr - ref).max() / max(np.abs(ref).max(), 1e-30))
    out.append(f"out={worst[0]:.1e} tgt={worst[1]:.1e}")
    print("  ".join(out), flush=True)


for dtype in ("f64", "mixed", "f32"):
    rirA, rirB = T.synth_rirs(200, 8, 16, 1)
    ap, orc, got, exp = T.run_pair(256, 128, rirA, rirB, 12, 2, 5, 4, 1.0, hops=6, dtype=dtype)
    report(f"small/{dtype}", ap, orc, got, exp, 129, 8, 16, (0, 1))
    ap.close()
rirA, rirB = T.cfg3_rirs()
for dtype, V, rA, rB in [("f64", 1, True, True), ("f64", 8, True, True), ("f64", 8, False, True), ("mixed", 1, True, True), ("f32", 8, True, True)]:
    t0 = time.time()
    x = T.pink(6 * 1024, 2024)
    ap, orc, got, exp = T.run_pair(2048, 1024, rirA, rirB, 16, 3, 7, V, 1.0, hops=6, run_A=rA, run_B=rB, dtype=dtype, x=x)
    report(f"cfg3/{dtype}/V{V}/{int(rA)}{int(rB)} ({time.time() - t0:.0f}s)", ap, orc, got, exp, 1025, 16, 32,
           tuple(z for z, r in enumerate((rA, rB)) if r))
    ap.close()
