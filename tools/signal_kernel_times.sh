#!/bin/bash
# On the GPU box: per-kernel durations of the cfg3 WHOLE-SIGNAL path (process_signal, float64) under rocprofv3, summed per chunk of
# sixteen hops: what a chunk's period is made of.   tools/signal_kernel_times.sh  ->  gpurun_out/signal_kernel_times.txt
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_sig
cat > /tmp/sig_run.py <<PY
import sys
sys.path.insert(0, "$REPO")
import numpy as np, bench
from ap_vast_unofficial_amd.apvast import apvast
N, H, L3, M3 = 2048, 1024, 16, 32
rng = np.random.default_rng(3)
env = np.exp(-np.arange(800) / 200.0)[:, None, None]
rirA = rng.standard_normal((800, L3, M3)) * env * 1e-3          # (taps, loudspeakers, microphones), as bench.also_cfg3
rirB = rng.standard_normal((800, L3, M3)) * env * 1e-3
hops = 320
x = rng.standard_normal((2, hops * H))
obj = apvast(N, rirA, rirB, 100, 20, 0, 0, 1, 1.0, 4 * N, hop_size=H, sampling_rate=48000, perceptual=False, dtype="f64", seed=0, device=0)
obj.process_signal(x[0, :32 * H], x[1, :32 * H])
obj.process_signal(x[0], x[1])
obj.close()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sig -- python3 /tmp/sig_run.py > /tmp/sig_run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob('/tmp/prof_sig/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the second call: 320 hops = 20 chunks; take the kernels after the first call's last kernel (a gap of host time)
t = [int(r["Start_Timestamp"]) for r in rows]
gaps = sorted(((t[i + 1] - t[i], i) for i in range(len(t) - 1)), reverse=True)[:6]
cut = max(i for g, i in gaps if i < len(t) * 0.5 and i > len(t) * 0.03) + 1
sel = rows[cut:]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r["Kernel_Name"]
    n = n.replace("void (anonymous namespace)::", "").split("(")[0][:70]
    acc[n][0] += 1
    acc[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (max(int(r["End_Timestamp"]) for r in sel) - min(int(r["Start_Timestamp"]) for r in sel)) / 1e3
chunks = 20
print(f"second process_signal call: 320 hops = {chunks} chunks, {len(sel)} kernels, span {span / 1e3:.3f} ms = {span / 320:.1f} us per hop = {span / chunks:.0f} us per chunk")
print("per chunk: launches | sum of kernel durations us (a kernel's duration includes the time it shares the chip) | kernel")
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{c / chunks:8.1f} | {d / chunks:9.1f} | {n}")
PY
