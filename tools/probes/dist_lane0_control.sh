run() { APV_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$1 timeout -k 10 300 python bench.py --steps 60 --no-also --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['blocks_resident_per_step'], d['config'].get('update_streams'), round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d.get('gather_check','')[:4])"; }
echo "1 lane"; run 29531; run 29532
echo "2 lanes, own control stream"; run 29533 --update-streams 2; run 29534 --update-streams 2
export APV_LANE0_CONTROL=1
echo "2 lanes, lane 0 = control stream"; run 29535 --update-streams 2; run 29536 --update-streams 2
echo "2 lanes, lane 0 = control, 8 blocks"; run 29537 --update-streams 2 --blocks 8
unset APV_LANE0_CONTROL
echo "1 lane, 8 blocks"; run 29538 --blocks 8
echo "single GPU path, lane 0 = control"; APV_LANE0_CONTROL=1 python bench.py --no-also --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"
