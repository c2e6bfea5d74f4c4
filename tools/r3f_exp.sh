#!/bin/bash
# tools/r3f_exp.sh -- round-3 experiments (development aid): 2-bit input tests, where the end-to-end traceback time goes
set -e
O=gpurun_out/r3f; mkdir -p $O
python3 -m pytest tests/test_gpu_packed2.py -x -q > $O/pytest_packed2.txt 2>&1 || { tail -n 40 $O/pytest_packed2.txt; exit 1; }
tools/e2e.sh 10000 long > $O/e2e_long.txt 2>&1
E2E_EXTRA="-pack2" tools/e2e.sh 10000 long > $O/e2e_long_pack2.txt 2>&1
E2E_EXTRA="-pool-gb 12" tools/e2e.sh 10000 long > $O/e2e_long_pool12.txt 2>&1
E2E_EXTRA="-pack2" tools/e2e.sh 100000 short > $O/e2e_short_pack2.txt 2>&1
tools/e2e.sh 100000 short > $O/e2e_short.txt 2>&1
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
dpx.write_pairs_file(dpx.make_batch(10000, 1024, 1024, seed=1), "/tmp/e2e_pairs.txt")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_lsw -o e2e -- $GRAFT_REPO_ROOT/dpx_gpu_genomics_project_amd/hostcpp/dpx_main -pairs /tmp/e2e_pairs.txt -algo LSW -match 3 -mismatch -1 -open -2 > /tmp/out_lsw.txt 2> $GRAFT_REPO_ROOT/$O/prof_lsw.log
grep -E "^Elapsed|^Kernel|^Traceback" /tmp/out_lsw.txt > $GRAFT_REPO_ROOT/$O/prof_lsw_times.txt
