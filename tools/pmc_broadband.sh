#!/bin/bash
# SQ counters of the broadband hop's kernels, per kernel name (two --pmc passes, no trace domains): tools/pmc_broadband.sh [reftest]
# -> gpurun_out/pmc_bb/counters.md
V=${1:-cfg1}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_bb; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
export APV_NO_GRAPH=1
if [ "$V" = "reftest" ]; then ARGS="3 reftest"; else ARGS="6"; fi
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/${V}_a -- python3 $REPO/tools/bench_broadband.py $ARGS > $OUT/${V}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/${V}_b -- python3 $REPO/tools/bench_broadband.py $ARGS > $OUT/${V}_b.log 2>&1
python3 - "$OUT" "$V" <<'PY'
import csv, glob, collections, sys, re
out, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in "ab":
    for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, v, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            n = re.sub(r"^void ", "", n).split("(")[0]
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU",
        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"]
lines = ["# SQ counters of the broadband hop's kernels (%s), per-dispatch means (tools/pmc_broadband.sh)" % v, "",
         "| kernel | dispatches | " + " | ".join(cols) + " | VALU instr / wave | wait-inst share of wave cycles | bank-conflict share of LDS-active |", "|---|---|" + "---|" * (len(cols) + 3)]
for n, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    m = {k: (sum(x) / len(x)) for k, x in c.items()}
    nd = len(c.get("SQ_WAVES", []))
    if nd == 0 or m.get("SQ_WAVES", 0) == 0:
        continue
    per_wave = m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"]
    wait = m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0
    bank = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else 0
    lines.append("| `%s` | %d | " % (n[:60], nd) + " | ".join("%.3g" % m.get(k, float("nan")) for k in cols) + " | %.0f | %.0f %% | %.0f %% |" % (per_wave, 100 * wait, 100 * bank))
open("%s/counters_%s.md" % (out, v), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
PY
