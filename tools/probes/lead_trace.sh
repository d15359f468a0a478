#!/bin/bash
# per-kernel times of the broadband hop at the reference's test parameters (n = 800) under rocprofv3, and the passes of the leading solver
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/lead_trace; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
APV_LEAD_DEBUG=1 python3 $REPO/tools/bench_broadband.py 3 reftest > $OUT/ref_debug.json 2> $OUT/ref_debug.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref -- python3 $REPO/tools/bench_broadband.py 4 reftest > $OUT/ref.json 2> $OUT/ref.err
f=$(ls -t $(find $OUT/ref -name "*kernel_stats.csv") | head -1); head -16 "$f" | cut -c1-60,120-260
grep -E "pass|degree|batch=2" $OUT/ref_debug.err | tail -40 | cut -c1-200
