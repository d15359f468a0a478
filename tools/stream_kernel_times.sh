#!/bin/bash
# On the GPU box: per-kernel mean durations of the cfg3 stream (float64, per-hop path) under rocprofv3.  tools/stream_kernel_times.sh [dtype]
DT=${1:-f64}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_sk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sk -- python3 $REPO/tools/bench_stream.py --hops 100 --dtype $DT > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_sk/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>5s}  {r['Name'][:90]}")
PY
