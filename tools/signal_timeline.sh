#!/bin/bash
# On the GPU box: kernel timeline of apvast.process_signal (cfg3 shape) under rocprofv3 --kernel-trace: one steady-state chunk of
# 16 hops (front half of the NEXT chunk beside it) and the period between consecutive joint-diagonalisation launches.
# tools/signal_timeline.sh [dtype] [hops]
DT=${1:-f64}
HOPS=${2:-96}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -- python3 $REPO/tools/bench_stream.py --hops $HOPS --dtype $DT --signal > /tmp/prof_tl.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/prof_tl/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Stream_Id', r.get('Queue_Id', '?')), r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0][:28]))
for f in glob.glob('/tmp/prof_tl/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'dma', 'copy ' + r.get('Direction', '')[:24]))
rows.sort()
g = [i for i, r in enumerate(rows) if r[3].startswith('gevd16m')]
if len(g) > 40:
    per = [(rows[g[i + 1]][0] - rows[g[i]][0]) / 1e3 for i in range(len(g) - 33, len(g) - 1)]
    print(f"period between GEVD launches, last 32 hops: mean {sum(per)/len(per):.1f} us (min {min(per):.1f}, max {max(per):.1f})")
    dur = [(rows[i][1] - rows[i][0]) / 1e3 for i in g[-33:-1]]
    print(f"GEVD kernel duration over the same hops: mean {sum(dur)/len(dur):.1f} us (min {min(dur):.1f}, max {max(dur):.1f})")
    i0 = g[-33]; t0 = rows[i0][0]
    print("\n| start us | end us | dur us | stream | kernel |\n|---:|---:|---:|---|---|")
    for r in rows[i0:g[-15] + 1]:
        print(f"| {(r[0]-t0)/1e3:.1f} | {(r[1]-t0)/1e3:.1f} | {(r[1]-r[0])/1e3:.1f} | {r[2]} | {r[3]} |")
PY
tail -2 /tmp/prof_tl.log
