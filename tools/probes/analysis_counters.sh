#!/bin/bash
# What bounds the chunk-wide analysis transforms and K1 of the whole-signal path (cfg3, float64)?  Separate --pmc passes of
# tools/bench_stream.py --signal: HBM bytes, LDS conflicts, issue stalls per kernel.   -> gpurun_out/analysis_counters/summary.md
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/analysis_counters; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
ARGS="--hops 128 --dtype f64 --signal"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_stream.py $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/tools/bench_stream.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/tools/bench_stream.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python3 $REPO/tools/bench_stream.py $ARGS > /dev/null 2>&1
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["OUT"]
dur = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)): acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
L = ["# whole-signal path, cfg3 shape, float64: per-kernel counters (means per dispatch; FETCH_SIZE x 2 x 1024 B, WRITE_SIZE x 1024 B)", "",
     "| kernel | calls | avg us | fetch MB | write MB | VALU-active % of wave cycles | issue-stall % | LDS conflict % of LDS-active |", "|---|---|---|---|---|---|---|---|"]
for k, c in sorted(acc.items(), key=lambda kv: -dur.get(kv[0], (0, 0))[0] * dur.get(kv[0], (0, 0))[1]):
    m = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else float("nan")
    calls, us = dur.get(k, (0, float("nan")))
    wc = m("SQ_WAVE_CYCLES")
    L.append("| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |" % (k[:90], calls, us, m("FETCH_SIZE") * 2048 / 1e6, m("WRITE_SIZE") * 1024 / 1e6,
             100 * m("SQ_ACTIVE_INST_VALU") / wc, 100 * m("SQ_WAIT_INST_ANY") / wc, 100 * m("SQ_LDS_BANK_CONFLICT") / max(m("SQ_LDS_IDX_ACTIVE"), 1)))
open(out + "/summary.md", "w").write("\n".join(L) + "\n")
print("\n".join(L))
PY
