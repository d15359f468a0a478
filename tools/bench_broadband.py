#!/usr/bin/env python3
"""cfg1 of BASELINE.json in the reference's own (broadband) formulation: bundled rirs.mat (8 x 8), N=256, J=32,
S=512, V=8: ms per hop on the GPU next to the CPU oracle (a float64 NumPy restatement of apvast.py)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    hops = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    g = np.load(os.path.join(ROOT, "tests", "golden", "rirs_cfg1.npz"))
    rirA, rirB = g["rirA"], g["rirB"]
    from ap_vast_unofficial_amd.apvast import apvast
    if len(sys.argv) > 2 and sys.argv[2] == "reftest":
        return reftest(hops, rirA, rirB)
    ap = apvast(256, rirA, rirB, 32, 16, 0, 0, 8, 1.0, 512, hop_size=128, perceptual=False, mode="broadband", seed=0)
    x = np.random.default_rng(7).standard_normal((2, (hops + 2) * 128))
    for h in range(2):
        ap.process_input_buffers(x[0, h * 128:(h + 1) * 128], x[1, h * 128:(h + 1) * 128])
    t0 = time.perf_counter()
    for h in range(2, hops + 2):
        ap.process_input_buffers(x[0, h * 128:(h + 1) * 128], x[1, h * 128:(h + 1) * 128])
    gpu = (time.perf_counter() - t0) / hops
    # the same hops through ONE call (apv_bb_process_signal: the joint diagonalisations of up to 8 hops as one batch)
    nsig = max(hops, 64)
    xs = np.random.default_rng(8).standard_normal((2, nsig * 128))
    ap.process_signal(xs[0, :16 * 128], xs[1, :16 * 128])
    t0 = time.perf_counter()
    ap.process_signal(xs[0], xs[1])
    gpu_sig = (time.perf_counter() - t0) / nsig
    from oracle.broadband import BroadbandOracle
    np.random.seed(0)
    orc = BroadbandOracle(256, rirA, rirB, 32, 16, 0, 0, 8, 1.0, 512, hop_size=128)
    for h in range(2):
        orc.process_input_buffers(x[0, h * 128:(h + 1) * 128], x[1, h * 128:(h + 1) * 128])
    t0 = time.perf_counter()
    nc = min(hops, 6)
    for h in range(2, nc + 2):
        orc.process_input_buffers(x[0, h * 128:(h + 1) * 128], x[1, h * 128:(h + 1) * 128])
    cpu = (time.perf_counter() - t0) / nc
    print(json.dumps({"workload": "cfg1 broadband 8x8, N=256, J=32 (n=256), S=512, V=8, both zones", "gpu_ms_per_hop": gpu * 1e3,
                      "gpu_ms_per_hop_process_signal": gpu_sig * 1e3, "process_signal_hops": nsig, "realtime_hop_ms": 128 / 48.0,
                      "cpu_oracle_ms_per_hop": cpu * 1e3, "cpu_count": os.cpu_count(), "speedup": cpu / gpu}))
def reftest(hops, rirA, rirB):
    """The reference's own test parameters (Python/make_python_test.m:6-15): block 1600, J = 100 (n = 800), V = 50,
    S = 1000, hop 800, 8 loudspeakers; first 8 microphones of the bundled RIRs."""
    from ap_vast_unofficial_amd.apvast import apvast
    from oracle.broadband import BroadbandOracle
    N, J, V, S, H = 1600, 100, 50, 1000, 800
    ap = apvast(N, rirA, rirB, J, 20, 6, 6, V, 1.0, S, perceptual=False, mode="broadband", seed=0)
    x = np.random.default_rng(7).standard_normal((2, (hops + 2) * H))
    for h in range(2):
        ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    t0 = time.perf_counter()
    for h in range(2, hops + 2):
        ap.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    gpu = (time.perf_counter() - t0) / hops
    nsig = max(hops, 8)
    xs = np.random.default_rng(8).standard_normal((2, nsig * H))
    ap.process_signal(xs[0], xs[1])                     # allocates the group buffers
    t0 = time.perf_counter()
    ap.process_signal(xs[0], xs[1])
    gpu_sig = (time.perf_counter() - t0) / nsig
    np.random.seed(0)
    orc = BroadbandOracle(N, rirA, rirB, J, 20, 6, 6, V, 1.0, S)
    for h in range(2):
        orc.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    t0 = time.perf_counter()
    orc.process_input_buffers(x[0, 2 * H:3 * H], x[1, 2 * H:3 * H])
    cpu = time.perf_counter() - t0
    print(json.dumps({"workload": "make_python_test.m parameters: 8x8, N=1600, J=100 (n=800), S=1000, V=50, both zones",
                      "gpu_ms_per_hop": gpu * 1e3, "gpu_ms_per_hop_process_signal": gpu_sig * 1e3, "process_signal_hops": nsig,
                      "realtime_hop_ms": H / 48.0, "cpu_oracle_ms_per_hop": cpu * 1e3, "cpu_count": os.cpu_count(),
                      "speedup": cpu / gpu}))


main()
