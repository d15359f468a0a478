#!/bin/bash
# rocprofv3 report for BASELINE config 5 (64 x 128 x 2048): correlation kernels and the full update.
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_cfg5; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_corr.py > $OUT/bench_corr.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/tools/bench_corr.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/tools/bench_corr.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_upd -- python3 $REPO/tools/bench_cfg5.py > $OUT/bench_cfg5.json 2>/dev/null
python3 - <<PY
import csv, glob, collections, json
out="$OUT"
def stats(sub):
    rows=[]
    for f in glob.glob(out+"/"+sub+"/**/*kernel_stats.csv", recursive=True):
        rows+=list(csv.DictReader(open(f)))
    return rows
def pmc(sub):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out+"/"+sub+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)): acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
L=["# rocprofv3 report, BASELINE config 5 (64 loudspeakers x 128 control points x 2048 bins)","",
   "## correlation stage alone (tools/bench_corr.py)","","| kernel | calls | avg us |","|---|---|---|"]
for r in stats("trace"):
    L.append("| \`%s\` | %s | %.1f |" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3))
f, w = pmc("fetch"), pmc("write")
L+=["","HBM traffic per launch (FETCH_SIZE x 1024 x 2 on gfx950, WRITE_SIZE x 1024):","","| kernel | fetch MB | write MB |","|---|---|---|"]
for k in f:
    fs=sum(f[k]["FETCH_SIZE"])/len(f[k]["FETCH_SIZE"])*1024*2/1e6
    ws=(sum(w[k]["WRITE_SIZE"])/len(w[k]["WRITE_SIZE"])*1024/1e6) if k in w else float("nan")
    L.append("| \`%s\` | %.1f | %.1f |" % (k[:110], fs, ws))
L+=["","algorithmic bytes: f32 2 x 134.2 MB in + 134.2 MB + 1 MB out; bf16 2 x 67.1 MB in + 134.2 MB out","",
    "bench line:","","\`\`\`",open(out+"/bench_corr.json").read().strip(),"\`\`\`","",
    "## full update (tools/bench_cfg5.py)","","| kernel | calls | avg us |","|---|---|---|"]
for r in stats("trace_upd"):
    L.append("| \`%s\` | %s | %.1f |" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3))
L+=["","\`\`\`",open(out+"/bench_cfg5.json").read().strip(),"\`\`\`"]
open(out+"/cfg5_roofline.md","w").write("\n".join(L)+"\n")
print("\n".join(L[:30]))
PY
