#!/bin/bash
# Runs on the GPU box: the records profiles/r04* and DESIGN.md quote.  tools/collect_r04.sh <tag>
TAG=${1:-r04a}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/round_$TAG; mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -c 300 $OUT/bench_$TAG.json; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $OUT/bench_20steps.json 2>/dev/null
bash tools/bb_kernel_times.sh > $OUT/bbk.log 2>&1
for v in cfg1 ref; do f=$(find gpurun_out/bb_kernels/$v -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/bb_kernel_stats_$v.csv; cp gpurun_out/bb_kernels/$v.json $OUT/bb_$v.json; done
APV_BB_TIMING=1 python tools/bench_broadband.py 6 > $OUT/bb_stage_times_cfg1.json 2> $OUT/bb_stage_times_cfg1.txt
APV_BB_TIMING=1 python tools/bench_broadband.py 4 reftest > $OUT/bb_stage_times_ref.json 2> $OUT/bb_stage_times_ref.txt
bash tools/profile_gpu.sh $TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
