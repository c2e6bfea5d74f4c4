#!/bin/bash
# tools/ab_ramp.sh -- ramp steps storing only the lines with cells (shipped) against whole ramp chunks (-DDPX_EXP_FULLRAMP=1,
# tools/bin/lib_fullramp.so), alternating on one box, bench.py timing (HIP events around every fill).
cd "$(dirname "$0")/.."
run() { DPX_LIB=$1 python3 bench.py --workload $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${1:-shipped}'.split('/')[-1], '$2', d['value'], 'GCUPS', d['roofline']['kernel_ms'], 'ms', 'frac', d['roofline']['frac'])"; }
for rep in 1 2 3; do
  for wl in ${WLS:-lsw_10k_1024 lnw_10k_1024 lsw_1k_512 anw_1k_1024}; do
    run tools/bin/lib_fullramp.so $wl
    run "" $wl
  done
done
