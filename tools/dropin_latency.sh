#!/bin/bash
# tools/dropin_latency.sh -- the reference's own main.cpp on the engine (one pair per align() call, 20 pthreads) vs the
# batched driver, 4000 short-read pairs (development aid)
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
dpx.write_pairs_file(make_ragged_batch(4000, 80, 130, 100, 160, seed=6), "/tmp/p4000.txt")
PY
make -s -C dpx_gpu_genomics_project_amd/hostcpp
for exe in oracle/_ref/main_dropin_LNW "dpx_gpu_genomics_project_amd/hostcpp/dpx_class_main -algo LNW" "dpx_gpu_genomics_project_amd/hostcpp/dpx_main -algo LNW"; do
  echo "== $exe"
  $exe -pairs /tmp/p4000.txt -match 3 -mismatch -1 -open -2 > /tmp/dropin_out.txt 2>/tmp/dropin_err.txt
  grep -E "Elapsed" /tmp/dropin_out.txt
done
echo "== reference CPU classes (oracle/_ref/ref_driver_O2 align mode)"
oracle/_ref/ref_driver_O2 time LNW /tmp/p4000.txt 3 -1 -2 -1 4000
echo "== reference CPU classes as the reference builds them (-O0)"
oracle/_ref/ref_driver time LNW /tmp/p4000.txt 3 -1 -2 -1 4000
echo "== phase trace of one thread's first pairs (DPX_TRACE=1)"
head -c 3000 /tmp/p4000.txt > /dev/null
python - <<PY
lines = open("/tmp/p4000.txt").read().split("\n")
open("/tmp/p400.txt", "w").write("\n".join(lines[:1200]) + "\n")
PY
DPX_TRACE=1 dpx_gpu_genomics_project_amd/hostcpp/dpx_class_main -algo LNW -pairs /tmp/p400.txt -match 3 -mismatch -1 -open -2 2>&1 >/dev/null | tail -16
