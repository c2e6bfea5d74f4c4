#!/bin/bash
# tools/ab_env.sh <rounds> <workload> "<ENV=.. ENV=..>" "<ENV=..>" ... -- alternate environment variants of ONE build on one box
# (fresh process per run); prints GCUPS, kernel ms and the pool record of every run.  Development aid.
ROUNDS=$1; WL=$2; shift 2
for i in $(seq 1 $ROUNDS); do
  for V in "$@"; do
    printf "%-44s " "[$V]"
    env $V python3 bench.py --workload $WL --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=r.get('pool') or {}
print(d['value'], r['kernel_ms'], 'frac', r['frac'], p.get('mode'), p.get('chunk_mb'), p.get('candidates_fill_ms'), p.get('candidates_memset_ms'), p.get('memset_tbps'))"
  done
done
