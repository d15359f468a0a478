#!/bin/bash
# What the chunk-wide analysis transforms of the whole-signal path wait for: the kernel's time with its stores, its loads or its
# stages taken out (APV_STFT_DEBUG = 1, 2, 4; results are wrong, only the time counts).  -> gpurun_out/analysis_bound.txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/analysis_bound; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
export APV_STFT_DEBUG_PROBE=1
for d in ${APV_PROBE_BITS:-0 1 2 4 3 7}; do
  export APV_STFT_DEBUG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d$d -- python3 $REPO/tools/bench_stream.py --hops 128 --dtype f64 --signal > /dev/null 2>&1
  f=$(find $OUT/d$d -name "*kernel_stats.csv" | head -1)
  echo "APV_STFT_DEBUG=$d: $(grep stft_analysis_jobs_kernel $f | cut -d, -f2-4,6,7 | tr -d '\"')  (calls, total ns, average ns, min, max)"
done
