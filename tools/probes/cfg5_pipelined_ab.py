#!/usr/bin/env python3
"""also.cfg5 of bench.py (two update streams, then one) from the package in the current directory: for A/B runs of two builds."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
from ap_vast_unofficial_amd import Engine
r = bench.also_cfg5(Engine, 0)
print(json.dumps({"where": os.getcwd().split("/")[-1], "pipelined_ms": round(r["ms_per_step"], 4), "alone_ms": round(r["roofline"]["kernel_ms"], 4), "value": round(r["value"])}), flush=True)
