#!/bin/bash
# per-kernel times of the broadband hop under rocprofv3 (plain launches: APV_NO_GRAPH=1), cfg1 and the reference's test parameters
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/bb_kernels; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
export APV_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg1 -- python3 $REPO/tools/bench_broadband.py 6 > $OUT/cfg1.json 2> $OUT/cfg1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ref -- python3 $REPO/tools/bench_broadband.py 3 reftest > $OUT/ref.json 2> $OUT/ref.err
for v in cfg1 ref; do
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; head -8 "$f" | cut -c1-200
done
