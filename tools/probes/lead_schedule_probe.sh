#!/bin/bash
# sweeps per pass of the leading solver's projected problem: schedules against the default (1 at b = 64, 2 at b = 32)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $REPO
for S in "" "2,1" "2,2,1" "3,1" "3,2,1" "1,2,1" "4,1" "2" ; do
  export APV_LEAD_SWEEP_SCHEDULE=$S; [ -z "$S" ] && unset APV_LEAD_SWEEP_SCHEDULE
  a=$(python tools/bench_broadband.py 40 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['gpu_ms_per_hop'])")
  b=$(python tools/bench_broadband.py 8 reftest | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['gpu_ms_per_hop'])")
  p=$(APV_LEAD_DEBUG=1 python tools/bench_broadband.py 3 reftest 2>&1 | grep "passes," | tail -1 | sed 's/.*: //')
  echo "schedule '${S:-default}': cfg1 $a ms/hop, n=800 $b ms/hop ($p)"
done
