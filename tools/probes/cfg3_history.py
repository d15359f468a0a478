#!/usr/bin/env python3
"""What in a process's history slows the chunked whole-signal path (cfg3_queue_phase.py: three engines created and closed before it
cost 40 %)?  mode: streams_destroyed | streams_alive | malloc | events, n = how many.  Raw HIP calls through ctypes."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode, n = sys.argv[1], int(sys.argv[2])
hip = C.CDLL("libamdhip64.so")
keep = []
if mode.startswith("streams"):
    ss = []
    for _ in range(n):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        ss.append(s)
    if mode == "streams_destroyed":
        for s in ss:
            hip.hipStreamDestroy(s)
    else:
        keep = ss
elif mode == "malloc":
    for _ in range(n):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(64 << 20)) == 0
        hip.hipFree(p)
import bench
a = bench.also_cfg3(0)
print(json.dumps({"mode": mode, "n": n, "cfg3_sig": round(a["process_signal"]["ms_per_hop"], 4), "runs": [round(r, 4) for r in a["process_signal"]["runs_ms_per_hop"]]}), flush=True)
