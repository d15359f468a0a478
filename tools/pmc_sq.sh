#!/bin/bash
# quick SQ-counter pass: tools/pmc_sq.sh <tag> [bench args]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for sub in "ab":
    acc=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%sub, recursive=True):
        for r in csv.DictReader(open(f)): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()): print("$TAG", k, "%.4g"%(sum(v)/len(v)))
PY
