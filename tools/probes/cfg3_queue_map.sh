#!/bin/bash
# which hardware queue does each kernel of the chunked whole-signal path run on, with 0 and with 3 streams created before it?
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/cfg3_queue_map; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for n in 0 3; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/pre$n -- python3 $REPO/tools/probes/cfg3_queue_phase.py $n > $OUT/pre$n.log 2>&1
  python3 - $OUT/pre$n $n <<'PY'
import csv, glob, sys, collections
d, n = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-20000:]          # the last process_signal call
m = collections.defaultdict(collections.Counter)
for r in tail:
    m[(r["Queue_Id"], r.get("Stream_Id", "?"))][r["Kernel_Name"].split("(")[0][-60:]] += 1
print(f"== {n} streams created before ==")
for k in sorted(m):
    print("queue %s stream %s: %s" % (k[0], k[1], ", ".join(f"{a} x{b}" for a, b in m[k].most_common(4))))
PY
  grep -h "cfg3_sig" $OUT/pre$n.log | cut -c1-160
done
