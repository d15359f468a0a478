#!/usr/bin/env python3
"""Probe: does the head / tail of a cfg2 launch (32 768 waves = 8 rounds of 4 per SIMD; 3.60 waves per SIMD alive on average,
profiles/r03/stage_stamps_32768_b.md) overlap with the next launch when consecutive steps go to DIFFERENT streams?
n engines (one stream each) share the device inputs; step i goes to engine i mod n, every engine into buffers of its own.
usage: two_stream_updates.py [blocks=32] [steps=200] [L=16 M=32]      (L = 64, M = 128, blocks = 2: BASELINE config 5)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ap_vast_unofficial_amd import Engine


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    K = blocks * 1024
    Lx = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    Mx = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    XB, XD, d = bench.synth(K, 1234, Lx, Mx)
    for n in (1, 2, 3, 1, 2):
        engs = [Engine(K, Lx, Mx, ranks=(Lx // 2,), mu=1.0, compute_dtype="f64", out_c128=False, device=0) for _ in range(n)]
        dXB, dXD, dd = engs[0].to_device(XB), engs[0].to_device(XD), engs[0].to_device(d)
        outs = [(e.alloc(K * Lx * 8), e.alloc(K * 4)) for e in engs]

        def run(count):
            for i in range(count):
                e = engs[i % n]
                e.update_dev(dXB, dXD, dd, outs[i % n][0], None, outs[i % n][1])

        def fence():
            for e in engs:
                e.sync()
            engs[0].device_sync()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            run(8 * n)
            fence()
        run(20)
        fence()
        t0 = time.perf_counter()
        run(steps)
        fence()
        dt = time.perf_counter() - t0
        st = outs[0][1].download((K,), np.int32)
        print(json.dumps({"streams": n, "L": Lx, "M": Mx, "blocks": blocks, "steps": steps, "ms_per_step": dt / steps * 1e3,
                          "updates_per_s": K * steps / dt, "status_nonzero": int(np.count_nonzero(st))}), flush=True)
        for b in (dXB, dXD, dd):
            b.free()
        for o in outs:
            o[0].free()
            o[1].free()
        for e in engs:
            e.close()


if __name__ == "__main__":
    main()
