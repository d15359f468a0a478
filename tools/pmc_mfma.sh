#!/bin/bash
# MFMA-busy counters (separate PMC passes, no trace options) for the correlation kernels of BASELINE config 5 and for
# the headline update kernel: tools/pmc_mfma.sh  ->  gpurun_out/pmc_mfma/mfma_busy.md
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_mfma; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for prog in bench_corr bench; do
  if [ $prog = bench ]; then CMD="$REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline"; else CMD="$REPO/tools/bench_corr.py"; fi
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${prog}_a -- python3 $CMD > $OUT/${prog}_a.log 2>&1 || echo "pass a failed for $prog"
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --output-format csv -d $OUT/${prog}_b -- python3 $CMD > $OUT/${prog}_b.log 2>&1 || echo "pass b failed for $prog"
done
python3 - <<PY
import csv, glob, collections
out="$OUT"
L=["# MFMA-busy counters (rocprofv3 --pmc, per-dispatch means)","",
   "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) x 256 CUs x 4 SIMDs): the rocprof derived metric (GRBM_GUI_ACTIVE is summed over the 8 XCDs)","",
   "| program | kernel | GRBM_GUI_ACTIVE | SQ_BUSY_CYCLES | SQ_VALU_MFMA_BUSY_CYCLES | MfmaUtil | MFMA MOPS f64 / f32 / bf16 | SQ_INSTS_MFMA |","|---|---|---|---|---|---|---|---|"]
for prog in ("bench_corr","bench"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in "ab":
        for f in glob.glob(f"{out}/{prog}_{sub}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)): acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in acc.items():
        if "rocclr" in k: continue
        m=lambda n: (sum(c[n])/len(c[n])) if c.get(n) else float("nan")
        util=m("SQ_VALU_MFMA_BUSY_CYCLES")/(m("GRBM_GUI_ACTIVE")/8*1024) if c.get("GRBM_GUI_ACTIVE") else float("nan")
        L.append("| %s | \`%s\` | %.4g | %.4g | %.4g | %.1f %% | %.4g / %.4g / %.4g | %.4g |" % (prog, k[:90], m("GRBM_GUI_ACTIVE"), m("SQ_BUSY_CYCLES"), m("SQ_VALU_MFMA_BUSY_CYCLES"), 100*util, m("SQ_INSTS_VALU_MFMA_MOPS_F64"), m("SQ_INSTS_VALU_MFMA_MOPS_F32"), m("SQ_INSTS_VALU_MFMA_MOPS_BF16"), m("SQ_INSTS_MFMA")))
open(out+"/mfma_busy.md","w").write("\n".join(L)+"\n")
print("\n".join(L))
PY
