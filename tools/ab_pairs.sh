#!/bin/bash
# usage: abp.sh <workload> <pairs> "<ENV..>" ...
WL=$1; P=$2; shift 2
for i in 1 2; do
  for V in "$@"; do
    printf "%-44s " "[$P $V]"
    env $V python3 bench.py --workload $WL --pairs $P --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], r['kernel_ms'], 'frac', r['frac'], d['config'].get('kernel'))"
  done
done
