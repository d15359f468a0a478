for s in 1 2 3 4; do
  echo "== APV_LEAD_SWEEPS=$s"
  APV_LEAD_SWEEPS=$s APV_LEAD_DEBUG=1 python tools/bench_broadband.py 6 2>&1 | grep -E "batch=2:|workload" | tail -3 | cut -c1-260
  APV_LEAD_SWEEPS=$s APV_LEAD_DEBUG=1 python tools/bench_broadband.py 4 reftest 2>&1 | grep -E "batch=2:|workload" | tail -3 | cut -c1-260
done
python tools/probes/gevd64_tune.py
