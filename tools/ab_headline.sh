#!/bin/bash
# tools/ab_headline.sh <libA.so> <libB.so> -- alternate two builds of libdpxalign.so on the headline bench on ONE box
# (boxes differ by a few %, so only same-box comparisons mean anything); development aid
A=$(realpath $1); B=$(realpath $2)
for i in 1 2 3; do
  for L in $A $B; do
    printf "%s " $(basename $L)
    DPX_LIB=$L python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
  done
done
