#!/usr/bin/env python3
"""Soak of the pipelined block updates (apv_set_update_streams): random sequences of launches into a few output buffers (some
shared between consecutive launches), unsynchronised input uploads and downloads in between, every download compared with the
one-stream result of the inputs then in place.  usage: update_lanes_soak.py [rounds=40]"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ap_vast_unofficial_amd import Engine

def cn(rng, *s):
    return ((rng.standard_normal(s) + 1j * rng.standard_normal(s)) * np.sqrt(0.5)).astype(np.complex64)

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(2026)
    K, L, M = 8192, 16, 32
    sets = [(cn(rng, K, M, L), cn(rng, K, M, L), cn(rng, K, M)) for _ in range(4)]
    ref_eng = Engine(K, L, M, ranks=(8,), mu=0.7, compute_dtype="f64", reg_dark=1e-7, out_c128=False)
    refs = [ref_eng.update(*s)[0] for s in sets]
    ref_eng.close()
    eng = Engine(K, L, M, ranks=(8,), mu=0.7, compute_dtype="f64", reg_dark=1e-7, out_c128=False)
    eng.set_update_streams(2)
    dev = [eng.to_device(a) for a in sets[0]]
    cur = 0
    outs = [eng.alloc(K * L * 8) for _ in range(3)]
    holds = [None] * 3               # which input set each output buffer was last computed from
    checks = launches = uploads = 0
    t0 = time.time()
    for r in range(rounds):
        for step in range(int(rng.integers(4, 12))):
            act = rng.random()
            if act < 0.2:                                    # new inputs, no host synchronisation behind the copies
                cur = int(rng.integers(0, len(sets)))
                for buf, a in zip(dev, sets[cur]):
                    eng._chk(eng.lib.apv_memcpy_h2d(eng.h, buf.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
                uploads += 1
            elif act < 0.85:                                 # a launch into a random buffer (often the one just written)
                o = int(rng.integers(0, 3))
                eng.update_dev(dev[0], dev[1], dev[2], outs[o])
                holds[o] = cur
                launches += 1
            else:                                            # a download in the middle of the sequence
                o = int(rng.integers(0, 3))
                if holds[o] is not None:
                    assert np.array_equal(outs[o].download((K, 1, L), np.complex64), refs[holds[o]]), (r, step, o)
                    checks += 1
        for o in range(3):
            if holds[o] is not None:
                assert np.array_equal(outs[o].download((K, 1, L), np.complex64), refs[holds[o]]), (r, "end", o)
                checks += 1
    eng.close()
    print(json.dumps({"rounds": rounds, "launches": launches, "uploads": uploads, "downloads_checked": checks, "all_equal": True,
                      "seconds": round(time.time() - t0, 1)}))

if __name__ == "__main__":
    main()
