#!/bin/bash
# usage: ab_lib.sh <rounds> <workload> <lib> ...   ("" = shipped)
R=$1; WL=$2; shift 2
for i in $(seq 1 $R); do for L in "$@"; do printf "%-28s " "[${L:-shipped}]"; DPX_LIB=$L python3 bench.py --workload $WL --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel_ms'], 'frac', r['frac'])"; done; done
