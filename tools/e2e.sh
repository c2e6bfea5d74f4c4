#!/bin/bash
# tools/e2e.sh -- end-to-end timing of the batched driver on the headline shape (development aid)
set -e
N=${1:-10000}
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
sb = dpx.make_batch($N, 1024, 1024, seed=1)
dpx.write_pairs_file(sb, "/tmp/e2e_pairs.txt")
PY
make -s -C dpx_gpu_genomics_project_amd/hostcpp
for algo in LSW LNW ANW; do
  EXT=""; OPEN=-2; [ $algo = ANW ] && EXT="-extend -1" && OPEN=-3
  echo "== $algo $N pairs, print to file"
  dpx_gpu_genomics_project_amd/hostcpp/dpx_main -pairs /tmp/e2e_pairs.txt -algo $algo -match 3 -mismatch -1 -open $OPEN $EXT -batch 5000 > /tmp/e2e_out.txt
  tail -9 /tmp/e2e_out.txt | grep -E "Elapsed|Kernel|Memory|Backtracking|Printing|GCUPS"
  ls -la /tmp/e2e_out.txt | awk '{print "output bytes", $5}'
done
