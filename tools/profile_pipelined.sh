#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the DEFAULT bench.py command (two update streams: launches overlap) and
# what the trace says about the overlap -> profiles/<tag>_pipelined.md.  The counters and the duration of a launch ALONE come from
# tools/profile_gpu.sh <tag> --update-streams 1.
# Usage: tools/profile_pipelined.sh <tag>
set -e
TAG=${1:-r04g}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_pipelined
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 100 --warmup 20 --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
python3 - "$OUT" "$TAG" "$REPO" "$ARGS" <<'PY'
import csv, glob, os, sys
out, tag, repo, args = sys.argv[1:5]
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gevd16m" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# the timed region of the command: the last 100 + 108 launches are the pipelined region and the one-stream leg behind it
n_alone, n_timed = 108, 100
alone = rows[-n_alone:][8:]
timed = rows[-(n_alone + n_timed):-n_alone]
def stats(rs):
    dur = [(e - s) / 1e3 for s, e, *_ in rs]
    span = (max(e for _, e, *_ in rs) - min(s for s, *_ in rs)) / 1e3
    both = 0.0
    for (s0, e0, *_), (s1, e1, *_) in zip(rs, rs[1:]):
        both += max(0, min(e0, e1) - max(s0, s1)) / 1e3
    return sum(dur) / len(dur), min(dur), max(dur), span / len(rs), both / span
a, t = stats(alone), stats(timed)
bench = [l for l in open(os.path.join(out, "trace.log")) if l.startswith("{")]
md = [f"# rocprofv3 kernel trace of the default (pipelined) command `{tag}`", "", f"command: `python3 bench.py {args}` (two update streams: the default on the single-GPU path)", "",
      "| launches | n | avg duration us | min | max | span / launches us | share of the span with two launches in flight |", "|---|---|---|---|---|---|---|",
      f"| timed region (launches alternate between two streams) | {len(timed)} | {t[0]:.1f} | {t[1]:.1f} | {t[2]:.1f} | {t[3]:.1f} | {100 * t[4]:.0f} % |",
      f"| one-stream leg behind it (`roofline.kernel_ms`) | {len(alone)} | {a[0]:.1f} | {a[1]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[4]:.0f} % |", "",
      "A launch of the timed region is in flight about twice as long as a launch alone, and two are in flight at any time: the rate is span / launches.",
      "queues seen in the timed region: " + ", ".join(sorted({f"queue {q} / stream {s}" for *_, q, s in timed})), "",
      "## bench line under the profiler", "", "```", bench[-1].strip() if bench else "(none)", "```", ""]
open(os.path.join(repo, "profiles", f"{tag}_pipelined.md"), "w").write("\n".join(md))
print("\n".join(md[:12]))
PY
mkdir -p "$REPO/gpurun_out/profiles_out" && cp "$REPO"/profiles/${TAG}_pipelined.md "$REPO/gpurun_out/profiles_out/"
