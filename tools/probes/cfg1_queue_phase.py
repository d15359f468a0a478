#!/usr/bin/env python3
"""The broadband whole-signal path (cfg1) after n streams were created in the process (see cfg3_queue_phase.py).
usage: cfg1_queue_phase.py <streams created and destroyed first> [ref]     (APV_BB_FRONT2 / APV_BB_FRONT_THREAD from the environment;
`ref`: the reference's test parameters, n = 800, instead of cfg1)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ap_vast_unofficial_amd import Engine
n_pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
engs = [Engine(64, 16, 32, ranks=(8,), mu=1.0, compute_dtype="f64", out_c128=False, device=0) for _ in range(n_pre)]
for e in engs:
    e.close()
ref = len(sys.argv) > 2 and sys.argv[2] == "ref"
a = bench.also_reftest(0) if ref else bench.also_cfg1(0)
print(json.dumps({"workload": "n800" if ref else "cfg1", "streams_before": n_pre, "front2": os.environ.get("APV_BB_FRONT2", "1"), "thread": os.environ.get("APV_BB_FRONT_THREAD", "1"),
                  "pib": round(a["process_input_buffers"]["ms_per_hop"], 4), "sig": round(a["process_signal"]["ms_per_hop"], 4),
                  "sig_out": round(a["process_signal_out"]["ms_per_hop"], 4)}), flush=True)
