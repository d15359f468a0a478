#!/bin/bash
# world-1 rehearsal of the distributed bench path (update lanes + RCCL all-gather) with the runtime's default number of hardware
# queues and with eight: a stream that shares a hardware queue with one that carries an event wait queues up behind that wait
set -e
run() { APV_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$1 timeout -k 10 300 python bench.py --steps 100 --no-also --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config'].get('update_streams'), d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('collective_us'), d.get('gather_check'))"; }
echo "default queues, 2 lanes"; run 29511
echo "default queues, 1 lane"; run 29512 --update-streams 1
echo "default queues, ring 2, 2 lanes"; APV_BENCH_RING=2 run 29516
export GPU_MAX_HW_QUEUES=8
echo "8 queues, 2 lanes"; run 29513
echo "8 queues, 1 lane"; run 29514 --update-streams 1
echo "8 queues, ring 2, 2 lanes"; APV_BENCH_RING=2 run 29515
echo "8 queues, single GPU path, 2 lanes"
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
