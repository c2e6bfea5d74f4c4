#!/bin/bash
# tools/e2e_ab.sh <algo> "<ENV=..>" "<ENV=..>" ... -- alternate environment variants of the batched driver on 10000 pairs of 1024 x 1024
# on one box (three rounds; development aid)
ALGO=$1; shift
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
dpx.write_pairs_file(dpx.make_batch(10000, 1024, 1024, seed=1), "/tmp/e2e_pairs.txt")
PY
EXT=""; OPEN=-2; [ $ALGO = ANW ] && EXT="-extend -1" && OPEN=-3
for i in 1 2 3; do
  for V in "$@"; do
    printf "%-28s " "[$ALGO $V]"
    env $V dpx_gpu_genomics_project_amd/hostcpp/dpx_main -pairs /tmp/e2e_pairs.txt -algo $ALGO -match 3 -mismatch -1 -open $OPEN $EXT > /tmp/e2e_out.txt
    grep -E "^Elapsed|^Kernel|^Memory|^Traceback" /tmp/e2e_out.txt | tr '\n' ' '; echo
  done
done
