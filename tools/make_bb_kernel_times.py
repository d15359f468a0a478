#!/usr/bin/env python3
"""profiles/r04/bb_kernel_times.txt from what tools/collect_r04.sh left in gpurun_out/round_<tag>/: the per-kernel statistics of the
broadband hop under rocprofv3 (cfg1 and the reference's test parameters), launches per call of the solver, the stage timer's lines."""
import csv, json, os, re, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "round_" + tag)
LEAD = ("lead_",)
PRE = ("chol_panel", "gemm64", "mirror_lower", "diag_inverse", "load_pair", "transpose_kernel", "coef_kernel", "vast_prefix", "tri_inverse", "gemm_kernel", "symmetrise")
JAC = ("block_jacobi", "la_fused", "la_solve")


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


out = ["# Broadband hop under rocprofv3 (tools/bb_kernel_times.sh via tools/collect_r04.sh %s; this file: tools/make_bb_kernel_times.py)" % tag, "",
       "Same script and arguments as profiles/r03/bb_kernel_times.txt: `tools/bench_broadband.py 6` (cfg1: 8 per-hop calls + process_signal over 16 and over 64 hops = 13 calls of apv_gevd_large)",
       "and `tools/bench_broadband.py 3 reftest` (the reference's test parameters: 5 per-hop calls + two process_signal calls over 8 hops = 7 calls).  Plain launches, no graphs on this path.",
       "Round 3, same script: cfg1 la_fused_kernel 3277 launches x 17.3 us = 56.5 ms (62 % of the trace), ~260 dependent launches per call; n = 800: 9797 x 19.2 us = 188 ms (76 %), ~750 per call.", ""]
for v, title in (("cfg1", "cfg1 (n = 256, V = 8)"), ("ref", "reference test parameters (n = 800, V = 50)")):
    rows = list(csv.DictReader(open(os.path.join(src, "bb_kernel_stats_%s.csv" % v))))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    grp = {"lead": [0, 0], "pre": [0, 0], "jacobi": [0, 0]}
    for r in rows:
        n = r["Name"]
        for key, pats in (("lead", LEAD), ("pre", PRE), ("jacobi", JAC)):
            if any(p in n for p in pats):
                grp[key][0] += int(r["Calls"])
                grp[key][1] += int(r["TotalDurationNs"])
    out.append("== " + title)
    out.append("all kernels: %.2f ms; leading solver %d launches, %.2f ms = %.1f %%; factorisation + whitening + filters %d launches, %.2f ms = %.1f %%; block Jacobi %d launches"
               % (tot / 1e6, grp["lead"][0], grp["lead"][1] / 1e6, 100.0 * grp["lead"][1] / tot, grp["pre"][0], grp["pre"][1] / 1e6,
                  100.0 * grp["pre"][1] / tot, grp["jacobi"][0]))
    out.append("")
    out.append('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for r in rows[:26]:
        out.append('"%s",%s,%s,%d,%s,%s,%s' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], round(float(r["AverageNs"])), r["Percentage"], r["MinNs"], r["MaxNs"]))
    out.append("")
    j = json.load(open(os.path.join(src, "bb_%s.json" % v)))
    out.append("under the profiler: %.3f ms per per-hop call, %.3f ms per hop through process_signal" % (j["gpu_ms_per_hop"], j["gpu_ms_per_hop_process_signal"]))
    out.append("")
out.append("Stage times of a per-hop call without the profiler (APV_BB_TIMING=1: synchronised at the stage boundaries), ms:")
for v in ("cfg1", "ref"):
    lines = [l.strip() for l in open(os.path.join(src, "bb_stage_times_%s.txt" % v)) if "batch=2:" in l or l.startswith("[apv bb]")]
    out.extend(lines[-2:])
out.append("")
out.append("(Earlier records of the round: git history of this file.)")
open(os.path.join(root, "profiles", "r04", "bb_kernel_times.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
