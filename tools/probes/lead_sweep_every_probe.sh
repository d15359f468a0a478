# does the Jacobi sweep have to run in every pass of the leading solver?  APV_LEAD_SWEEP_EVERY=k: sweeps in passes 0, 1 and every k-th
for e in 1 2 3; do
  echo "== APV_LEAD_SWEEP_EVERY=$e"
  APV_LEAD_SWEEP_EVERY=$e APV_LEAD_DEBUG=1 python tools/bench_broadband.py 6 2>&1 | grep -E "batch=2:|workload" | tail -3 | cut -c1-230
  APV_LEAD_SWEEP_EVERY=$e APV_LEAD_DEBUG=1 python tools/bench_broadband.py 4 reftest 2>&1 | grep -E "batch=2:|workload" | tail -3 | cut -c1-230
done
