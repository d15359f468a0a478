#!/usr/bin/env python3
"""also.cfg3 of bench.py from the package in the current directory: for A/B runs of two builds (e.g. back streams of the chunked path)."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
a = bench.also_cfg3(0)
print(json.dumps({"where": os.getcwd().split("/")[-1], "cfg3_pib": round(a["process_input_buffers"]["ms_per_hop"], 4), "cfg3_sig": round(a["process_signal"]["ms_per_hop"], 4),
                  "runs": [round(r, 4) for r in a["process_signal"].get("runs_ms_per_hop", [])], "equal": a["values_checked"]["process_signal_equals_hop_loop"]}), flush=True)
