#!/bin/bash
# tools/ab_sizes.sh <libA.so> <libB.so> -- two builds of the library across batch sizes and workloads on ONE box (development aid)
A=$(realpath $1); B=$(realpath $2)
run() { printf "%-14s %-22s " "$(basename $1)" "$2 $3"; DPX_LIB=$1 python bench.py --workload $2 ${3:+--pairs $3} --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"; }
for rep in 1 2; do
  for spec in "lsw_10k_1024 10000" "lsw_10k_1024 5000" "lsw_10k_1024 7000" "lsw_10k_1024 2500" "lsw_1k_512" "anw_1k_1024" "lnw_100k_short" "bsw_10k_4096_b128 2000"; do
    run $A $spec; run $B $spec
  done
done
