#!/bin/bash
# tools/tb_kernel.sh -- kernel durations of the traceback kernels alone (rocprofv3 kernel trace of tools/tb_kernel.py; TB_SHAPES, DPX_TB_WALK)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tbk && rocprofv3 --kernel-trace -d /tmp/tbk -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/tb_kernel.py > /tmp/tbk.log 2>&1
python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("/tmp/tbk/t_kernel_trace.csv"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cur, acc = None, []
def flush():
    if cur: print(f"{cur[0]:24s} grid {cur[1]:>8s}  n={len(acc):2d}  min {min(acc):8.1f} us  median {sorted(acc)[len(acc)//2]:8.1f} us")
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if not name.startswith("k_traceback"): continue
    key = (name, r["Grid_Size_X"])
    if key != cur: flush(); cur, acc = key, []
    acc.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
flush()
PY
