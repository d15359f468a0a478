#!/bin/bash
# grouped control-point spectra [K/4][C][4] (default where the order-16 float64 kernel reads them) against bin-major (APV_SPECTRA_GROUP=1):
# per-kernel times of the cfg3 whole-signal path and of the per-hop path under rocprofv3
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$REPO/gpurun_out/spectra_group; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for g in ${APV_PROBE_GROUPS:-4 1}; do
  export APV_SPECTRA_GROUP=$g
  for mode in --signal ""; do
    tag=g${g}_$([ -n "$mode" ] && echo signal || echo hops)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $REPO/tools/bench_stream.py --hops 128 --dtype f64 $mode > $OUT/$tag.json 2>/dev/null
    f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
    echo "== group $g $([ -n "$mode" ] && echo process_signal || echo hop loop): $(cut -c1-160 $OUT/$tag.json)"
    python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("stft_analysis_jobs","gevd16m","fir_fft_kernel","istft_ola","apply_filters")):
        print("   %-60s calls %5s avg %8.1f us  max %8.1f" % (r["Name"].replace("(anonymous namespace)::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, int(r["MaxNs"])/1e3))
PY
  done
done
