#!/bin/bash
# tools/e2e.sh [N] [short] -- end-to-end timing of the batched driver (development aid): N pairs of the headline shape
# (1024x1024), or with "short" of the reference's own dataset shape (reference 100-160, query 80-130)
set -e
N=${1:-10000}
SHAPE=${2:-long}
BATCHARG=""; [ -n "${3:-}" ] && BATCHARG="-batch $3"   # default: dpx_main sizes its batches from its pool budget (-pool-gb 4)
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
sb = make_ragged_batch($N, 80, 130, 100, 160, seed=6) if "$SHAPE" == "short" else dpx.make_batch($N, 1024, 1024, seed=1)
dpx.write_pairs_file(sb, "/tmp/e2e_pairs.txt")
PY
make -s -C dpx_gpu_genomics_project_amd/hostcpp
for algo in ${ALGOS:-LSW LNW ANW}; do
  EXT=""; OPEN=-2; [ $algo = ANW ] && EXT="-extend -1" && OPEN=-3
  echo "== $algo $N pairs ($SHAPE), print to file"
  dpx_gpu_genomics_project_amd/hostcpp/dpx_main -pairs /tmp/e2e_pairs.txt -algo $algo -match 3 -mismatch -1 -open $OPEN $EXT $BATCHARG ${E2E_EXTRA:-} > /tmp/e2e_out.txt
  grep -E "^Elapsed|^Kernel|^Memory|^Backtracking|^Traceback|^Printing|^GCUPS" /tmp/e2e_out.txt | tr '\n' ' '; echo
  ls -la /tmp/e2e_out.txt | awk '{print "output bytes", $5}'
done
