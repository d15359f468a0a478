#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 rocpd database (APV_NO_GRAPH=1 run of jdiag_large_probe.py): average
duration of the block rounds, split into the first round of a sweep (full inner sweep) and the others."""
import collections, re, sqlite3, statistics, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
agg = collections.defaultdict(list)
for n, s, e in rows:
    m = re.search(r"(\w+_kernel)", n)
    agg[m.group(1) if m else n[:40]].append((e - s) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:34s} n={len(v):5d} avg={sum(v)/len(v):8.2f} us  total={sum(v)/1e3:8.3f} ms")
rounds_per_sweep = int(sys.argv[2]) if len(sys.argv) > 2 else 15
jr = [(s, e) for n, s, e in rows if "block_jacobi" in n]
d = [(e - s) / 1e3 for s, e in jr]
full = [x for i, x in enumerate(d) if i % rounds_per_sweep == 0]
cross = [x for i, x in enumerate(d) if i % rounds_per_sweep != 0]
if full and cross:
    f, c = statistics.median(full), statistics.median(cross)
    inner = (f - c) / 15
    print(f"full round {f:.2f} us, cross round {c:.2f} us -> inner round {inner:.3f} us, rest of a round {c - 16 * inner:.2f} us")
