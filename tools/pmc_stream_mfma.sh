#!/bin/bash
# On the GPU box: MFMA busy fraction per kernel of the cfg3 stream (float64).  tools/pmc_stream_mfma.sh
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_sm
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d /tmp/pmc_sm -- python3 $REPO/tools/bench_stream.py --hops 40 --dtype f64 > /tmp/pmc_sm.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_sm/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get('GRBM_GUI_ACTIVE', 0) <= 0: continue
    util = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (m['GRBM_GUI_ACTIVE'] / 8 * 256 * 4)
    print(f"{k:60s} MfmaUtil {100*util:5.1f} %  gui_active/8 {m['GRBM_GUI_ACTIVE']/8:.3e} cycles  MFMA insts {m.get('SQ_INSTS_MFMA',0):.3e}  VALU {m.get('SQ_INSTS_VALU',0):.3e}")
PY
