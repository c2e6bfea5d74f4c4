#!/bin/bash
# tools/tb_kernel.sh -- kernel durations of the traceback / text kernels alone (rocprofv3 kernel trace of tools/tb_kernel.py)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tbk && rocprofv3 --kernel-trace -d /tmp/tbk -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/tb_kernel.py 2>/dev/null
python3 - <<PY
import csv, collections
d = collections.defaultdict(list)
order = []
for r in csv.DictReader(open("/tmp/tbk/t_kernel_trace.csv")):
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if name.startswith("k_traceback") or name.startswith("k_out"):
        key = (name, r["Grid_Size_X"])
        if key not in d: order.append(key)
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in order:
    v = d[k]
    print(f"{k[0]:28s} grid {k[1]:>8s}  n={len(v):3d}  min {min(v):8.1f} us  median {sorted(v)[len(v)//2]:8.1f} us")
PY
