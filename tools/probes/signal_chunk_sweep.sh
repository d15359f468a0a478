#!/bin/bash
# cfg3 process_signal ms/hop against the chunk size of the whole-signal path (APV_SIGNAL_CHUNK), two runs each
for c in 8 16 24 32 48; do
  for rep in 1 2; do
    echo -n "chunk=$c: "; APV_SIGNAL_CHUNK=$c python3 tools/bench_stream.py --signal 2>&1 | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_hop'])"
  done
done
