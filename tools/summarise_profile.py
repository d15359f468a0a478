#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (rocprofv3 CSVs) into profiles/<tag>_summary.md
and profiles/traffic.json (HBM bytes per launch of the dominant kernel, gfx950-corrected)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
args = sys.argv[3:]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(repo, "profiles")
os.makedirs(prof, exist_ok=True)


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out_dir, sub, "**", pat), recursive=True))


lines = [f"# rocprofv3 summary `{tag}`", "", f"command: `python3 bench.py {' '.join(args)}`", ""]
kern = defaultdict(list)
for f in find("trace", "*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        kern[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
lines += ["## kernel trace (per-kernel durations, us)", "", "| kernel | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|"]
dominant = None
for name, d in sorted(kern.items(), key=lambda kv: -sum(kv[1])):
    short = name if len(name) < 110 else name[:107] + "..."
    lines.append(f"| `{short}` | {len(d)} | {sum(d)/len(d):.1f} | {min(d):.1f} | {max(d):.1f} | {sum(d)/1e3:.3f} |")
    if dominant is None:
        dominant = name
for f in find("trace", "*kernel_stats.csv"):
    lines += ["", "### rocprofv3 --stats (verbatim)", "", "```"] + open(f).read().strip().splitlines()[:12] + ["```"]


def pmc(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(sub, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


traffic = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    acc = pmc(sub)
    if not acc:
        continue
    lines += ["", f"## {sub} (per-dispatch mean)", "", "| kernel | counter | mean | dispatches |", "|---|---|---|---|"]
    for name, cs in acc.items():
        short = name if len(name) < 80 else name[:77] + "..."
        for c, v in cs.items():
            lines.append(f"| `{short}` | {c} | {sum(v)/len(v):.4g} | {len(v)} |")
            if name == dominant:
                traffic[c] = sum(v) / len(v)

if "FETCH_SIZE" in traffic or "WRITE_SIZE" in traffic:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the
    # bytes of a wide coalesced streaming read -> x2 on the read side.  WRITE_SIZE is exact.  The guide calibrates that factor
    # for 16-byte-per-lane reads; profiles/r03/fetch_calibration.md measures it for this kernel's own slab reads (8 bytes per lane,
    # a 4 KB slab per wave) against a known 1 GiB: 2.000 as well.
    factor = 2.0
    try:
        cal = json.load(open(os.path.join(prof, "r03", "fetch_calibration.json")))
        factor = float(cal["kernels"]["slab8"]["factor"])
    except Exception:
        pass
    fetch = traffic.get("FETCH_SIZE", 0.0) * 1024 * factor
    write = traffic.get("WRITE_SIZE", 0.0) * 1024
    blocks, dtype = 32, "f64"
    for i, a in enumerate(args):
        if a == "--blocks":
            blocks = int(args[i + 1])
        if a == "--dtype":
            dtype = args[i + 1]
    tj = {"tag": tag, "kernel": dominant, "blocks": blocks, "dtype": dtype, "fetch_factor": factor,
          "fetch_factor_source": "profiles/r03/fetch_calibration.md (1 GiB read in this kernel's slab pattern)",
          "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
          "raw": traffic}
    json.dump(tj, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
    lines += ["", "## HBM traffic of the dominant kernel (per launch)", "",
              f"FETCH_SIZE x 1024 x {factor:.3f} (gfx950 correction, calibrated on this access pattern: profiles/r03/fetch_calibration.md) = {fetch:.4g} B; WRITE_SIZE x 1024 = {write:.4g} B; "
              f"total {fetch + write:.4g} B"]
for log in ("trace.log",):
    p = os.path.join(out_dir, log)
    if os.path.exists(p):
        js = [l for l in open(p) if l.startswith("{")]
        if js:
            lines += ["", "## bench line under the profiler", "", "```", js[-1].strip(), "```"]
open(os.path.join(prof, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
