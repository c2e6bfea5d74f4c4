#!/bin/bash
# tools/trace_e2e.sh -- phase timings (DPX_TRACE=1) of the batched driver on the short-read shape (development aid)
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
dpx.write_pairs_file(make_ragged_batch(100000, 80, 130, 100, 160, seed=6), "/tmp/e2e_pairs.txt")
PY
make -s -C dpx_gpu_genomics_project_amd/hostcpp
DPX_TRACE=1 dpx_gpu_genomics_project_amd/hostcpp/dpx_main -pairs /tmp/e2e_pairs.txt -algo LNW -match 3 -mismatch -1 -open -2 -batch 20000 2>&1 >/tmp/e2e_out.txt | head -80
grep -E "Elapsed|Kernel time|Memory man|Backtr|Printing" /tmp/e2e_out.txt
