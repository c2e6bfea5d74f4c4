#!/bin/bash
# tools/bsw_e2e.sh [N] [LEN] [BAND] -- end-to-end timing of the batched driver on banded SW (development aid)
set -e
N=${1:-4000}; LEN=${2:-4096}; BAND=${3:-128}
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
dpx.write_pairs_file(dpx.make_batch($N, $LEN, $LEN, seed=4), "/tmp/bsw_pairs.txt")
PY
make -s -C dpx_gpu_genomics_project_amd/hostcpp
M=dpx_gpu_genomics_project_amd/hostcpp/dpx_main
for w in ${WALKS:-2 0}; do
  for i in 1 2; do
    echo "== BSW $N pairs of $LEN x $LEN, band $BAND, DPX_TB_WALK=$w"
    DPX_TB_WALK=$w $M -pairs /tmp/bsw_pairs.txt -algo BSW -band $BAND -match 3 -mismatch -1 -open -2 > /tmp/bsw_out_$w.txt
    grep -E "^Elapsed|^Kernel|^Back|^Trace|^Printing|^Memory|^GCUPS" /tmp/bsw_out_$w.txt | tr "\n" " "; echo
  done
done
grep -vE "time|GCUPS" /tmp/bsw_out_2.txt | md5sum; grep -vE "time|GCUPS" /tmp/bsw_out_0.txt | md5sum
