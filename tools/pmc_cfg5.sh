#!/bin/bash
# BASELINE config 5 (64 x 128 x 2048, float64): kernel trace and SQ / LDS / MFMA counters of the update kernel, for the
# order-64 kernel (kernels_gevd64.hip) and for the LDS kernel it replaced (APV_NO_GEVD64=1 selects it).
#   tools/pmc_cfg5.sh  ->  gpurun_out/pmc_cfg5/cfg5_counters.md
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_cfg5; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for v in ${APV_CFG5_VARIANTS:-new}; do
  if [ $v = old ]; then export APV_NO_GEVD64=1; else unset APV_NO_GEVD64; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_trace -- python3 $REPO/tools/bench_cfg5.py f64 > $OUT/${v}_bench.json 2> $OUT/${v}_trace.err
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/${v}_sq -- python3 $REPO/tools/bench_cfg5.py f64 > /dev/null 2> $OUT/${v}_sq.err || echo "SQ pass failed ($v)"
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${v}_sq2 -- python3 $REPO/tools/bench_cfg5.py f64 > /dev/null 2> $OUT/${v}_sq2.err || echo "SQ2 pass failed ($v)"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${v}_fetch -- python3 $REPO/tools/bench_cfg5.py f64 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${v}_write -- python3 $REPO/tools/bench_cfg5.py f64 > /dev/null 2>&1
done
unset APV_NO_GEVD64
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["OUT"]
L = ["# BASELINE config 5 (64 loudspeakers x 128 control points x 2048 bins, float64): the update kernel under rocprofv3", "",
     "`new` = gevd64_kernel (kernels_gevd64.hip), `old` = gevd_vast_kernel<double, 64, 1024, ...> (kernels_gevd.hip, APV_NO_GEVD64=1).",
     "Counters are per-dispatch means of separate --pmc passes (no trace options in those passes).", ""]
for v in os.environ.get("APV_CFG5_VARIANTS", "new").split():
    L += [f"## {v}", "", "| kernel | calls | avg us |", "|---|---|---|"]
    for f in glob.glob(f"{out}/{v}_trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rocclr" in r["Name"]: continue
            L.append("| `%s` | %s | %.1f |" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("sq", "sq2", "fetch", "write"):
        for f in glob.glob(f"{out}/{v}_{sub}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if "rocclr" in k or "gevd" not in k: continue
        m = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else float("nan")
        L += ["", "`%s`" % k[:100], "", "| counter | per-dispatch mean |", "|---|---|"]
        for n in sorted(c): L.append("| %s | %.5g |" % (n, m(n)))
        wc = m("SQ_WAVE_CYCLES")
        L += ["",
              "- VALU-active share of wave cycles: %.1f %%; issue-stall (SQ_WAIT_INST_ANY) share: %.1f %%" % (100 * m("SQ_ACTIVE_INST_VALU") / wc, 100 * m("SQ_WAIT_INST_ANY") / wc),
              "- LDS: %.4g instructions, %.4g active cycles, %.4g of them bank conflicts (%.1f %%)" % (m("SQ_INSTS_LDS"), m("SQ_LDS_IDX_ACTIVE"), m("SQ_LDS_BANK_CONFLICT"), 100 * m("SQ_LDS_BANK_CONFLICT") / max(m("SQ_LDS_IDX_ACTIVE"), 1)),
              "- MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): %.1f %%" % (100 * m("SQ_VALU_MFMA_BUSY_CYCLES") / (m("GRBM_GUI_ACTIVE") / 8 * 1024)),
              "- HBM traffic per launch: FETCH_SIZE x 1024 x 2 = %.4g B, WRITE_SIZE x 1024 = %.4g B (algorithmic: 2048 x 132 608 = 2.716e8 B)" % (m("FETCH_SIZE") * 2048, m("WRITE_SIZE") * 1024)]
    L += ["", "bench line:", "", "```", open(f"{out}/{v}_bench.json").read().strip(), "```", ""]
open(out + "/cfg5_counters.md", "w").write("\n".join(L) + "\n")
# HBM bytes per launch of the order-64 kernel for bench.py's also.cfg5 record (profiles/traffic_cfg5.json)
import json
for f in glob.glob(f"{out}/new_fetch/**/*counter_collection.csv", recursive=True): pass
acc = collections.defaultdict(list)
for sub in ("fetch", "write"):
    for f in glob.glob(f"{out}/new_{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gevd64" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc.get("FETCH_SIZE") and acc.get("WRITE_SIZE"):
    fe = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"]) * 1024 * 2
    wr = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"]) * 1024
    json.dump({"kernel": "gevd64x2_kernel<fused>", "updates_per_launch": 2048, "dtype": "f64", "fetch_bytes_corrected_x2": fe, "write_bytes": wr,
               "hbm_bytes_per_launch": fe + wr, "algorithmic_bytes_per_launch": 2048 * 133632,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of tools/bench_cfg5.py f64; FETCH_SIZE x 2 per MI355X_MICROARCH.md (calibrated: profiles/r03/fetch_calibration.md); the scratch slots of the bins in flight are L2 / Infinity-Cache traffic that these counters include"},
              open(out + "/traffic_cfg5.json", "w"), indent=1)
print("\n".join(L))
PY
