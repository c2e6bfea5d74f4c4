#!/bin/bash
# tools/ab_quad.sh -- ablation of the short-read quad kernels on one box: the shipped library against builds without
# global stores (-DDPX_EXP_QUAD=1), without the LDS round trip (-DDPX_EXP_QUAD=2) and without the recurrence
# (-DDPX_EXP_NOCOMPUTE=1), alternating twice.  Build the variants into tools/bin/ first (see profiles/README.md).
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "" tools/bin/quad_exp1.so tools/bin/quad_exp2.so tools/bin/quad_nocompute.so; do
    for wl in ${WLS:-lnw_100k_short}; do
      DPX_LIB=$v python3 bench.py --workload $wl --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${v:-shipped}', '$wl', d['value'], 'GCUPS', d['roofline']['kernel_ms'], 'ms')"
    done
  done
done
