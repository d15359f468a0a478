#!/bin/bash
# world-1 rehearsal of the sharded bench path (one update stream, one RCCL all-gather per step) at 8, 32 and 64 blocks per rank and
# step: a launch boundary costs the same head and tail once per step, whatever the step holds
set -e
run() { APV_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$1 timeout -k 10 300 python bench.py --steps $2 --no-also --no-cpu-baseline --blocks $3 "${@:4}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['blocks_resident_per_step'], d['config'].get('update_streams'), d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('collective_us'), d.get('gather_check'))"; }
run 29521 100 8
run 29522 50 32
run 29523 30 64
run 29524 30 64 --update-streams 2
run 29525 100 8
