#!/usr/bin/env python3
"""Where do the ~37 us between a cfg3 hop's kernels (139 us) and a process_input_buffers call (0.176 ms) go?  The C call with a
fresh result array per call (what Engine.process_block does), with ONE reused array, with a page-locked array, and the class call
around it.  (Round 4: class 0.175-0.183, C call 0.170-0.178, reused array -1 us; a variant of the library that DMAs straight into a
page-locked caller array -- the copy taken out of the captured graph -- measured 0.1797 against 0.185 for pageable arrays on a box
where the shipped form measures 0.176-0.185: the two extra runtime calls cost what the host's pass over 512 KB saves.  Not kept.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from ap_vast_unofficial_amd.apvast import apvast
from ap_vast_unofficial_amd import _capi
N, H, L3, M3, P = 2048, 1024, 16, 32, 800
rng = np.random.default_rng(99)
env = np.exp(-np.arange(P) / 120.0)[:, None, None]
rirA = rng.standard_normal((P, L3, M3)) * env * 1e-3
rirB = rng.standard_normal((P, L3, M3)) * env * 1e-3
hops = 300
x = bench.pink(hops * H, 2024)
obj = apvast(N, rirA, rirB, 100, 20, 0, 0, 1, 1.0, 4 * N, hop_size=H, sampling_rate=48000, perceptual=False, dtype="f64", seed=0, device=0)
eng = obj._eng
for h in range(8):
    obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
t0 = time.perf_counter()
for h in range(hops):
    obj.process_input_buffers(x[0, h * H:(h + 1) * H], x[1, h * H:(h + 1) * H])
t_class = (time.perf_counter() - t0) / hops
n_out = obj._n_out
fn = eng.lib.apv_process_block_f64
ins = [(np.ascontiguousarray(x[0, h * H:(h + 1) * H]), np.ascontiguousarray(x[1, h * H:(h + 1) * H])) for h in range(hops)]
t0 = time.perf_counter()
for a, b in ins:
    out = np.empty((n_out // L3, H, L3))
    fn(eng.h, _capi._ptr(a), _capi._ptr(b), _capi._ptr(out))
t_fresh = (time.perf_counter() - t0) / hops
out = np.empty((n_out // L3, H, L3))
t0 = time.perf_counter()
for a, b in ins:
    fn(eng.h, _capi._ptr(a), _capi._ptr(b), _capi._ptr(out))
t_reuse = (time.perf_counter() - t0) / hops
pin = eng.pinned_empty((n_out // L3, H, L3))
t0 = time.perf_counter()
for a, b in ins:
    fn(eng.h, _capi._ptr(a), _capi._ptr(b), _capi._ptr(pin))
t_pin = (time.perf_counter() - t0) / hops
print(json.dumps({"class_call_ms": round(t_class * 1e3, 4), "c_call_fresh_array_ms": round(t_fresh * 1e3, 4), "c_call_reused_array_ms": round(t_reuse * 1e3, 4),
                  "c_call_page_locked_array_ms": round(t_pin * 1e3, 4), "result_bytes": int(out.nbytes)}))
obj.close()
