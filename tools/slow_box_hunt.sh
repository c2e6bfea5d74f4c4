#!/bin/bash
# tools/slow_box_hunt.sh -- development aid: the default bench on THIS box; if the box is one of the slow ones (the first pool
# candidate, 256-MiB chunks, fills in >= 3.6 ms) also the headline on other pool constructions, one fresh process each.
mkdir -p gpurun_out
OUT=gpurun_out/slow_box_hunt_$(date +%s).txt
run() { printf "%-34s " "[$1]"; env $1 python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=r.get('pool') or {}
print(d['value'], r['kernel_ms'], 'frac', r['frac'], p.get('candidates_kind'), p.get('candidates_fill_ms'), 'kept', p.get('kept'))"; }
{
  run "DPX_X=0"
  FIRST=$(python3 - <<PY
import re
t=open("$OUT").read() if False else ""
PY
)
} 2>&1 | tee $OUT
SLOW=$(python3 -c "
import re,sys
t=open('$OUT').read()
m=re.search(r'\] \[([0-9.]+),', t.split('frac')[1]) if 'frac' in t else None
print(1 if m and float(m.group(1)) >= 3.6 else 0)")
if [ "$SLOW" = 1 ]; then
  { echo "slow box: more constructions"; for V in "DPX_POOL_CHUNK_MB=512" "DPX_POOL_CHUNK_MB=1024" "DPX_POOL_CHUNK_MB=2048" "DPX_POOL_CHUNK_MB=4096" "DPX_POOL_CHUNK_MB=32768" "DPX_POOL_CHUNK_MB=64" "DPX_POOL=malloc" "DPX_X=0"; do run "$V"; done; } 2>&1 | tee -a $OUT
fi
