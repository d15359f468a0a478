#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/sig800; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
APV_LEAD_DEBUG=1 APV_BB_TIMING=1 python3 $REPO/tools/probes/bb_signal_n800.py 16 > $OUT/debug.out 2> $OUT/debug.err
cat $OUT/debug.out
sed -n '/timed call/,$p' $OUT/debug.err | cut -c1-220 | head -80
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $REPO/tools/probes/bb_signal_n800.py 16 > $OUT/prof.out 2> $OUT/prof.err
cat $OUT/prof.out
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -24 "$f" | cut -c1-70,120-260
