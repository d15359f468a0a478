#!/bin/bash
# Runs on the GPU box: everything the round's DESIGN.md / profiles/ quote.  tools/collect_round.sh <tag>
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/round_$TAG; mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench_f64.json 2> $OUT/bench_f64.err; tail -c 400 $OUT/bench_f64.json; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_f64_20steps.json 2>/dev/null      # the driver's own short run
python bench.py --dtype f32 --no-cpu-baseline > $OUT/bench_f32.json 2>/dev/null
APV_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 python bench.py --no-cpu-baseline > $OUT/bench_dist_rehearsal.json 2> $OUT/bench_dist.err
for dt in f64 mixed f32; do python tools/bench_stream.py --hops 468 --dtype $dt $([ $dt = f64 ] && echo --cpu-hops 2) > $OUT/stream_cfg3_$dt.json 2>/dev/null; cat $OUT/stream_cfg3_$dt.json; done
for dt in f64 mixed f32; do python tools/bench_stream.py --hops 468 --dtype $dt --signal > $OUT/stream_cfg3_signal_$dt.json 2>/dev/null; cat $OUT/stream_cfg3_signal_$dt.json; done
python tools/bench_broadband.py 20 > $OUT/broadband_cfg1.json 2>/dev/null; cat $OUT/broadband_cfg1.json
python tools/bench_broadband.py 5 reftest > $OUT/broadband_reftest.json 2>/dev/null; cat $OUT/broadband_reftest.json
python tools/bench_cfg5.py > $OUT/cfg5.json 2>/dev/null; cat $OUT/cfg5.json
export TMPDIR=/tmp; cd /tmp
for dt in f64 mixed; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stream_$dt -- python3 $REPO/tools/bench_stream.py --hops 100 --dtype $dt > /dev/null 2>&1
  cp $OUT/prof_stream_$dt/*/*kernel_stats.csv $OUT/stream_kernel_stats_$dt.csv; rm -rf $OUT/prof_stream_$dt
done
bash $REPO/tools/signal_timeline.sh f64 > $OUT/signal_timeline_f64.md 2>&1; head -1 $OUT/signal_timeline_f64.md
export TMPDIR=/tmp; cd /tmp
# the broadband path under the profiler WITH its replayed hipGraphs (this crashed rocprofv3 in round 1; see profiles/r02/rocprof_graph.md)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_jl -- python3 $REPO/tools/probes/jdiag_large_probe.py 2 > $OUT/jdiag_large_graph.log 2>&1; echo "jdiag_large under rocprofv3, graphs on: rc=$?" >> $OUT/jdiag_large_graph.log
cp $OUT/prof_jl/*/*kernel_stats.csv $OUT/gevd_large_kernel_stats.csv; rm -rf $OUT/prof_jl
rocprofv3 --kernel-trace --stats -d $OUT/prof_jl2 -o jl -- python3 $REPO/tools/probes/jdiag_large_probe.py 2 >> $OUT/jdiag_large_graph.log 2>&1; echo "same with -o jl (rocpd output, round 1's command line): rc=$?" >> $OUT/jdiag_large_graph.log; rm -rf $OUT/prof_jl2
tail -3 $OUT/jdiag_large_graph.log
cd $REPO
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
