#!/bin/bash
# tools/r3g_exp.sh -- round-3 experiments (development aid): the scalar wave traceback -- parity tests, times by path length, end to end
set -e
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_traceback.py tests/test_gpu_drivers.py -x -q > $O/pytest_traceback.txt 2>&1 || { tail -n 40 $O/pytest_traceback.txt; exit 1; }
for W in default 0 2; do
  echo "== DPX_TB_WALK=$W"
  if [ $W = default ]; then timeout -k 10 300 python3 tools/tb_time.py both; timeout -k 10 300 python3 tools/tb_time.py mid; else DPX_TB_WALK=$W timeout -k 10 300 python3 tools/tb_time.py both; DPX_TB_WALK=$W timeout -k 10 300 python3 tools/tb_time.py mid; fi
done > $O/tb_time.txt 2>&1
tools/e2e.sh 10000 long > $O/e2e_long.txt 2>&1
tools/e2e.sh 10000 long >> $O/e2e_long.txt 2>&1
python - <<PY
import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
dpx.write_pairs_file(dpx.make_batch(10000, 1024, 1024, seed=1), "/tmp/e2e_pairs.txt")
PY
DPX_TRACE=1 dpx_gpu_genomics_project_amd/hostcpp/dpx_main -pairs /tmp/e2e_pairs.txt -algo LSW -match 3 -mismatch -1 -open -2 > /tmp/out_lsw.txt 2> $O/trace_lsw.txt
grep -E "^Elapsed|^Kernel|^Traceback|^Backtracking|^Memory" /tmp/out_lsw.txt >> $O/trace_lsw.txt
