#!/bin/bash
# On the GPU box: FETCH_SIZE of three kernels that each read exactly 1 GiB (tools/probes/fetch_calib) -> the factor that turns the
# counter into bytes for the order-16 kernel's own access pattern.  Writes profiles/r03/fetch_calibration.json (+ .md).
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_fc
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fc -- $REPO/tools/probes/fetch_calib/fetch_calib > /tmp/pmc_fc.log 2>&1
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_fc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
nbytes = 1 << 30
out = {"bytes_per_kernel": nbytes, "unit_note": "FETCH_SIZE is in KiB (x 1024 -> bytes)", "kernels": {}}
lines = ["# FETCH_SIZE against a known byte count (tools/probes/fetch_calib, 1 GiB read once per kernel)", "",
         "| kernel | access | FETCH_SIZE x 1024 (mean of the last two of three launches) | bytes / counter |", "|---|---|---|---|"]
what = {"slab8": "8 B per lane, each wave its own 4 KB slab in 8 loads (the order-16 kernel's slab reads)",
        "stream16": "16 B per lane, streaming (the guide's calibrated case: factor 2)", "stream8": "8 B per lane, streaming"}
for k, v in sorted(acc.items()):
    m = sum(v[1:]) / max(1, len(v[1:])) * 1024
    name = k.strip()
    out["kernels"][name] = {"fetch_size_bytes": m, "factor": nbytes / m if m else None}
    lines.append(f"| {name} | {what.get(name, '')} | {m:.4g} | {nbytes / m if m else float('nan'):.3f} |")
import os
os.makedirs("$REPO/gpurun_out/r03g", exist_ok=True)
json.dump(out, open("$REPO/gpurun_out/r03g/fetch_calibration.json", "w"), indent=1)
open("$REPO/gpurun_out/r03g/fetch_calibration.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
