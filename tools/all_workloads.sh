#!/bin/bash
# tools/all_workloads.sh [outfile] -- one bench.py line per workload (no CPU baseline), on one box
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/all_workloads.jsonl}; : > $OUT
for wl in lsw_10k_1024 lnw_10k_1024 lsw_1k_512 anw_1k_1024 bsw_10k_4096_b128 lnw_100k_short lsw_100k_short anw_100k_short; do
  python3 bench.py --workload $wl --no-cpu-baseline 2>/dev/null >> $OUT
done
python3 - $OUT <<'PY'
import sys, json
for l in open(sys.argv[1]):
    d = json.loads(l); r = d["roofline"]
    print(f"{d['config']['algorithm']:4s} {d['config']['pairs_per_gpu']:7d} {d['config']['kernel']:18s} {d['value']:9.1f} GCUPS  {r['kernel_ms']:8.4f} ms  frac {r['frac']:.3f}" + (f"  in-band {r['in_band_kernel_gcups']}" if 'in_band_kernel_gcups' in r else ""))
PY
