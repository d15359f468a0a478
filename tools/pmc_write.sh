#!/bin/bash
# On the GPU box: WRITE_SIZE / FETCH_SIZE of the bench kernel (quick check for register spills).  tools/pmc_write.sh
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_w
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline > /tmp/pmc_w.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/pmc_w/*/*counter_collection.csv')[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == 'WRITE_SIZE' and 'gevd16m' in r['Kernel_Name']]
print(f"WRITE_SIZE mean {sum(v)/len(v):.1f} (x1024 B = {sum(v)/len(v)*1024/1e6:.1f} MB per launch) over {len(v)} dispatches")
PY
grep -o '"ms_per_step": [0-9.]*' /tmp/pmc_w.log | head -1
