#!/usr/bin/env python3
"""cfg3 of BASELINE.json: single-GPU streaming, 10 s pink noise at 48 kHz, N=2048, H=1024, 16 x 32,
end-to-end blocks/s including H2D of each hop and D2H of the (H, L) outputs."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pink(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((2, n))
    X = np.fft.rfft(x, axis=1)
    f = np.arange(X.shape[1], dtype=float)
    f[0] = 1.0
    X /= np.sqrt(f)
    X[:, 0] = 0.0
    y = np.fft.irfft(X, n, axis=1)
    return (y / np.sqrt((y ** 2).mean(axis=1, keepdims=True))).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hops", type=int, default=468)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--cpu-hops", type=int, default=0, help="also time the CPU oracle on this many hops")
    ap.add_argument("--L", type=int, default=16)
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--N", type=int, default=2048)
    ap.add_argument("--V", type=int, default=1)
    ap.add_argument("--signal", action="store_true", help="one process_signal call (hops pipelined) instead of the hop loop")
    args = ap.parse_args()
    from ap_vast_unofficial_amd.apvast import apvast
    N, H, L, M, P = args.N, args.N // 2, args.L, args.M, 800
    rng = np.random.default_rng(99)
    env = np.exp(-np.arange(P) / 120.0)[:, None, None]
    rirA = rng.standard_normal((P, L, M)) * env * 1e-3
    rirB = rng.standard_normal((P, L, M)) * env * 1e-3
    x = pink(args.hops * H, 2024)
    obj = apvast(N, rirA, rirB, 100, 20, 0, 0, args.V, 1.0, 4 * N, hop_size=H, sampling_rate=48000,
                 perceptual=False, dtype=args.dtype, seed=0)
    # timed through the CLASS (what a reference caller calls): process_input_buffers per hop, or one process_signal call
    for h in range(4):
        obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    if args.signal:
        # the caller's output array, touched once: a 10 s signal returns 184 MB and fresh pages would be timed otherwise
        sig_out = np.zeros(obj.signal_output_shape(args.hops * H), obj.signal_output_dtype)
        obj.process_signal(x[0, :32 * H], x[1, :32 * H])
    t0 = time.perf_counter()
    if args.signal:
        obj.process_signal(x[0], x[1], out=sig_out)
    else:
        for h in range(args.hops):
            obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
    dt = time.perf_counter() - t0
    out = {"entry": "apvast.process_signal" if args.signal else "apvast.process_input_buffers", "workload": f"cfg3 streaming N={N} H={H} L={L} M={M} V={args.V} rir_len={P}", "hops": args.hops,
           "blocks_per_s": args.hops / dt, "ms_per_hop": dt / args.hops * 1e3,
           "realtime_factor": (args.hops * H / 48000.0) / dt, "dtype": args.dtype,
           "subband_updates_per_s": args.hops * (N // 2 + 1) * 2 / dt}
    if args.cpu_hops:
        from oracle.subband_stream import SubbandStreamOracle
        orc = SubbandStreamOracle(N, rirA, rirB, 20, 0, 0, list(range(1, args.V + 1)), 1.0, hop_size=H)
        t0 = time.perf_counter()
        for h in range(args.cpu_hops):
            orc.process(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
        out["cpu_oracle_blocks_per_s"] = args.cpu_hops / (time.perf_counter() - t0)
        out["cpu_count"] = os.cpu_count()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
