#!/bin/bash
# tools/e2e_timeline.sh ALGO [min_us] -- device timeline (kernels + copies) of one dpx_main run on /tmp/e2e_pairs.txt (run tools/e2e.sh first)
ALGO=${1:-LSW}; MINUS=${2:-100}
EXT=""; OPEN=-2; [ $ALGO = ANW ] && EXT="-extend -1" && OPEN=-3
R=$GRAFT_REPO_ROOT; M=$R/dpx_gpu_genomics_project_amd/hostcpp/dpx_main
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o t --output-format csv -- $M -pairs /tmp/e2e_pairs.txt -algo $ALGO -match 3 -mismatch -1 -open $OPEN $EXT ${E2E_EXTRA:-} > /tmp/tl_out.txt 2>&1
grep -E "^Elapsed|^Kernel|^Backtracking|^Traceback" /tmp/tl_out.txt | tr '\n' ' '; echo
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40], "q" + r.get("Queue_Id", "")))
for f in glob.glob("/tmp/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") , ""))
rows.sort()
t0 = rows[0][0]
for s, e, name, q in rows:
    if (e - s) / 1e3 >= $MINUS: print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e3:9.1f} us {q:>4}  {name}")
PY
