#!/bin/bash
# tools/pmc.sh <outdir> <cmd...> -- run rocprofv3 PMC passes (one counter group per pass, kernel-trace only) on the GPU box.
# Usage on the box:  tools/pmc.sh gpurun_out/pmc1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
 "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES"
 "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_64B TCC_EA0_WRREQ"
 "GRBM_GUI_ACTIVE GRBM_TA_BUSY"
 "TCC_BUSY TCC_TAG_STALL TCC_NORMAL_WRITEBACK TCC_NORMAL_EVICT"
 "TCC_HIT TCC_MISS TCC_REQ TCC_READ"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
[ "${PMC_NO_TRAFFIC:-0}" = 1 ] && GROUPS_=("${GROUPS_[@]:0:6}")   # FETCH_SIZE / WRITE_SIZE come from tools/profile_all.sh
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1))
  ( cd "$REPO" && rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$OUT/g$i" -- "$@" > "$OUT/g$i.log" 2>&1 )
  echo "group $i rc=$?" >> "$OUT/status.txt"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rocclr" in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in acc.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"  {c:36s} mean {sum(v)/len(v):.6g}  (n={len(v)})\n")
print(open(out + "/summary.txt").read())
PY
