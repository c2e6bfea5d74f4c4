#!/bin/bash
# tools/box_survey.sh -- one block per GPU box of the pool (development aid): which device this is, what clkprobe sees, the
# headline / short-read / banded / affine fills of the shipped build, and what rocm-smi reports (shader clock, power, temperature)
# WHILE the headline fill runs in a loop.  Every gpurun call lands on a fresh box: run it several times and collect
# gpurun_out/box_survey_*.txt to see how far boxes differ for one binary, and whether the slow ones run the fill at a lower clock.
mkdir -p gpurun_out
OUT=gpurun_out/box_survey_$(date +%s).txt
ID=$(rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id" | head -1 | awk '{print $NF}')
[ -x tools/bin/clkprobe ] || hipcc --offload-arch=gfx950 -O2 tools/clkprobe.hip -o tools/bin/clkprobe 2>/dev/null
CLK=$(tools/bin/clkprobe 2>/dev/null | tail -1 | sed 's/clkprobe rep 2: //')
line() { python3 bench.py --workload $1 --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=r.get('pool') or {}
print('$1', d['value'], 'GCUPS', r['kernel_ms'], 'ms frac', r['frac'], 'pool_fill_ms', p.get('candidates_fill_ms'), 'memset_ms', p.get('candidates_memset_ms'))"; }
{
  echo "box $(hostname) gpu ${ID:-?} | $CLK"
  rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -E "Max Graphics Package Power|Performance Level" | head -2
  echo "-- idle:"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Graphics Package Power|junction" | head -4
  python3 bench.py --workload lsw_10k_1024 --no-cpu-baseline --steps 2500 > /tmp/bs_long.json 2>/dev/null &
  BP=$!
  sleep 5
  for k in 1 2 3; do echo "-- under the headline fill (sample $k):"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Graphics Package Power|junction" | head -4; sleep 1; done
  wait $BP
  python3 -c "
import json; d=json.load(open('/tmp/bs_long.json')); r=d['roofline']; print('lsw_10k_1024 x2500 fills', d['value'], 'GCUPS', r['kernel_ms'], 'ms frac', r['frac'])"
  line lsw_10k_1024; line lnw_100k_short; line bsw_10k_4096_b128; line anw_1k_1024
} 2>&1 | tee $OUT
