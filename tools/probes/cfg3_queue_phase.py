#!/usr/bin/env python3
"""Does the whole-signal path of the subband stream depend on how many streams the process has created before it?  (HIP deals
streams over GPU_MAX_HW_QUEUES hardware queues, four by default, in the order of their creation; the chunked path has six or
seven streams of its own, so some of them share a queue, and which ones do depends on what came before.)
usage: cfg3_queue_phase.py <streams created and destroyed first>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ap_vast_unofficial_amd import Engine
n_pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
engs = []
for i in range(n_pre):               # one stream per engine
    engs.append(Engine(64, 16, 32, ranks=(8,), mu=1.0, compute_dtype="f64", out_c128=False, device=0))
for e in engs:
    e.close()
a = bench.also_cfg3(0)
print(json.dumps({"streams_before": n_pre, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default"),
                  "cfg3_pib": round(a["process_input_buffers"]["ms_per_hop"], 4), "cfg3_sig": round(a["process_signal"]["ms_per_hop"], 4),
                  "runs": [round(r, 4) for r in a["process_signal"]["runs_ms_per_hop"]]}), flush=True)
