# hops per batched joint diagonalisation of the broadband whole-signal path (APV_BB_GROUP), cfg1 and the reference's test parameters
for g in 1 2 4 8 16; do echo "cfg1 G=$g"; APV_BB_GROUP=$g python tools/bench_broadband.py 4 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gpu_ms_per_hop_process_signal'])"; done
for g in 1 2 4 8 16; do echo "n=800 G=$g"; APV_BB_GROUP=$g python tools/bench_broadband.py 16 reftest 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gpu_ms_per_hop_process_signal'], d['gpu_ms_per_hop'])"; done
