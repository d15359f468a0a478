#!/bin/bash
# On the GPU box: VALU / LDS / MFMA instruction counts of the order-16 float64 kernel per stage (the kernel cut short by
# debug_stop, tools/probes/gevd16_stages.py), from one rocprofv3 --pmc pass.   tools/pmc_stages16.sh
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmc_st
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU --output-format csv -d /tmp/pmc_st -- python3 $REPO/tools/probes/gevd16_stages.py > /tmp/pmc_st.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_st/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
# dispatches come in six groups of 120 launches (20 warm-up + 100 timed) in the order of the probe's stage list
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(by)
names = ["correlate", "cholesky + inverse", "whitening", "float cholesky", "one-sided sweeps", "refinement + tail"]
per = len(ids) // len(names)
prev = collections.Counter()
print("| stage (cumulative kernel cut at its end) | VALU / wave | LDS / wave | MFMA / wave | SALU / wave | stage VALU | stage LDS | stage MFMA |")
print("|---|---|---|---|---|---|---|---|")
for g, name in enumerate(names):
    grp = ids[g * per:(g + 1) * per]
    mean = {c: sum(by[i].get(c, 0) for i in grp) / len(grp) / 32768 for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_SALU")}
    print(f"| {name} | {mean['SQ_INSTS_VALU']:.0f} | {mean['SQ_INSTS_LDS']:.0f} | {mean['SQ_INSTS_MFMA']:.0f} | {mean['SQ_INSTS_SALU']:.0f} | "
          f"{mean['SQ_INSTS_VALU'] - prev['SQ_INSTS_VALU']:.0f} | {mean['SQ_INSTS_LDS'] - prev['SQ_INSTS_LDS']:.0f} | {mean['SQ_INSTS_MFMA'] - prev['SQ_INSTS_MFMA']:.0f} |")
    prev = collections.Counter(mean)
PY
