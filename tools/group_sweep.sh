#!/bin/bash
# tools/group_sweep.sh -- headline fill by group size of the interleaved matrix layout (DPX_GROUP: how many launch-adjacent waves share
# one block of chunks) on THIS box (development aid): do the boxes on which the default is slow prefer another write pattern?
mkdir -p gpurun_out
OUT=gpurun_out/group_sweep_$(date +%s).txt
run() { printf "%-40s " "[$1]"; env $1 python3 bench.py --workload ${2:-lsw_10k_1024} --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=r.get('pool') or {}
print(d['value'], r['kernel_ms'], 'frac', r['frac'], p.get('candidates_fill_ms'), p.get('candidates_memset_ms'))"; }
{
  for V in "DPX_X=0" "DPX_GROUP=1" "DPX_GROUP=8" "DPX_GROUP=16" "DPX_GROUP=32" "DPX_GROUP=128" "DPX_GROUP=256" "DPX_GROUP=1024" "DPX_X=0" "DPX_POOL=malloc" "DPX_POOL_CHUNK_MB=2" "DPX_POOL_CHUNK_MB=1024"; do run "$V"; done
} 2>&1 | tee $OUT
