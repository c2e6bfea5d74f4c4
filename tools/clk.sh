#!/bin/bash
# tools/clk.sh "<name>:<lib>:<env>" ... -- shader cycles / effective clock / VALU instructions of the fill kernels per variant
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  IFS=: read v lib envs <<< "$spec"
  L=$R/tools/bin/libdpx_$lib.so; [ "$lib" = default ] && L=$R/dpx_gpu_genomics_project_amd/libdpxalign.so
  rm -rf /tmp/clk_$v
  env DPX_LIB=$L $envs rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/clk_$v -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline ${WORKLOAD:+--workload $WORKLOAD} > /tmp/clk_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
rows = [r for f in glob.glob(f"/tmp/clk_{v}/*/*_counter_collection.csv") for r in csv.DictReader(open(f)) if "fill" in r["Kernel_Name"]]
by = {}
for r in rows:
    d = by.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"][:60]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for d, c in sorted(by.items(), key=lambda x: int(x[0]))[-2:]:
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"{v:10s} {c['dur']:6.0f} us  {cyc/1e6:5.2f} Mcycles  {cyc/c['dur']/1e3:.2f} GHz  VALU {c['SQ_INSTS_VALU']/1e9:.2f}e9  LDS {c['SQ_INSTS_LDS']/1e6:.0f}e6  SALU {c['SQ_INSTS_SALU']/1e6:.0f}e6  waitinst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f}  wait {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f}  {c['name']}")
PY
done
