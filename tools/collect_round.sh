#!/bin/bash
# Runs on the GPU box: everything the round's DESIGN.md / profiles/ quote.  tools/collect_round.sh <tag>
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/round_$TAG; mkdir -p $OUT
cd $REPO
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_f64.json 2> $OUT/bench_f64.err; tail -c 600 $OUT/bench_f64.json
python bench.py --dtype f32 --no-cpu-baseline > $OUT/bench_f32.json 2>/dev/null
python tools/bench_stream.py --hops 468 --cpu-hops 2 > $OUT/stream_cfg3.json 2>/dev/null; cat $OUT/stream_cfg3.json
python tools/bench_broadband.py 20 > $OUT/broadband_cfg1.json 2>/dev/null; cat $OUT/broadband_cfg1.json
python tools/bench_broadband.py 5 reftest > $OUT/broadband_reftest.json 2>/dev/null; cat $OUT/broadband_reftest.json
python tools/bench_cfg5.py > $OUT/cfg5.json 2>/dev/null; cat $OUT/cfg5.json
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stream -- python3 $REPO/tools/bench_stream.py --hops 100 > /dev/null 2>&1
# (rocprofv3 cannot trace replayed hipGraphs of the broadband path: APV_NO_GRAPH=1 launches the rounds one by one)
APV_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d $OUT/prof_jl -o jl -- python3 $REPO/tools/probes/jdiag_large_probe.py 2 > /dev/null 2>&1
python3 $REPO/tools/probes/rounds_from_db.py $OUT/prof_jl/jl_results.db > $OUT/gevd_large_kernels.txt; tail -1 $OUT/gevd_large_kernels.txt; rm -rf $OUT/prof_jl
cp $OUT/prof_stream/*/*kernel_stats.csv $OUT/stream_kernel_stats.csv

rm -rf $OUT/prof_stream
