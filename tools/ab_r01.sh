#!/bin/bash
# tools/ab_r01.sh -- the shipped library against the round-1 build (tools/bin/lib_r01.so), alternating, on one box
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for wl in ${WLS:-anw_1k_1024 lsw_1k_512 lsw_10k_1024 lnw_10k_1024 bsw_10k_4096_b128 lnw_100k_short anw_100k_short}; do
    DPX_LIB=tools/bin/lib_r01.so python3 tools/ab_fill.py $wl
    python3 tools/ab_fill.py $wl
  done
done
