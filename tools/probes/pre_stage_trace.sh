#!/bin/bash
# factor-and-whiten stage of the broadband hop: kernel times under rocprofv3, new (recursive inverse, 64 x 64 products) against APV_LARGE_OLDPRE=1
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/pre_trace; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for v in new old; do
  [ $v = old ] && export APV_LARGE_OLDPRE=1
  for c in cfg1 ref; do
    a="6"; [ $c = ref ] && a="4 reftest"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_$c -- python3 $REPO/tools/bench_broadband.py $a > $OUT/${v}_$c.json 2> $OUT/${v}_$c.err
    f=$(find $OUT/${v}_$c -name "*kernel_stats.csv" | head -1)
    echo "== $v $c"; grep -E "chol_panel|tri_inverse|gemm|mirror|symmetrise|diag_inverse|transpose_kernel|load_pair" "$f" | cut -c1-70,100-200
    cut -c1-200 $OUT/${v}_$c.json
  done
done
