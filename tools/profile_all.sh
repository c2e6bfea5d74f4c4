#!/bin/bash
# tools/profile_all.sh <round> [workloads...] -- the evidence behind bench.py's roofline object for EVERY BASELINE workload, on
# the GPU box, into gpurun_out/profile_<round>/ (copy to profiles/<round>/):
#   <wl>_bench.json              bench.py line (with cpu_baseline for the headline)
#   <wl>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the same command (+ <wl>_bench_under_rocprof.json)
#   <wl>_pmc.json                WRITE_SIZE and FETCH_SIZE per fill, separate --pmc passes (KiB counters; FETCH_SIZE doubled for
#                                gfx950 per MI355X_MICROARCH.md)
# and pmc_traffic.json = {kernel_source_sha256, workloads: {wl: HBM bytes per fill}} which bench.py reports as roofline.traffic.
set -u
ROUND=$1; shift
WLS=${*:-lsw_10k_1024 lnw_10k_1024 lsw_1k_512 anw_1k_1024 bsw_10k_4096_b128 lnw_100k_short lsw_100k_short anw_100k_short}
REPO=$(pwd); OUT=$REPO/gpurun_out/profile_$ROUND; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $REPO
# which clock does this box hold under a VALU load / under a store load (boxes of the pool differ by up to 17 % on the latency-bound kernels)
[ -x tools/bin/clkprobe ] || hipcc --offload-arch=gfx950 -O2 tools/clkprobe.hip -o tools/bin/clkprobe 2>/dev/null
[ -x tools/bin/clkprobe ] && tools/bin/clkprobe > $OUT/clkprobe.txt 2>&1
for WL in $WLS; do
  EXTRA="--no-cpu-baseline"; [ $WL = lsw_10k_1024 ] && EXTRA=""
  python3 bench.py --workload $WL $EXTRA > $OUT/${WL}_bench.json 2> $OUT/${WL}_bench.err || { echo "bench $WL failed"; continue; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${WL}_stats -o st -- python3 bench.py --workload $WL --no-cpu-baseline \
      > $OUT/${WL}_bench_under_rocprof.json 2> $OUT/${WL}_stats.err || { echo "stats $WL failed"; continue; }
  cp $(find $OUT/${WL}_stats -name '*kernel_stats.csv' | head -1) $OUT/${WL}_kernel_stats.csv
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${WL}_$c -o pm -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline \
        > $OUT/${WL}_$c.log 2>&1 || echo "pmc $c $WL failed"
  done
  echo "done $WL"
done
python3 - $OUT $WLS <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
from bench import kernel_source_hash
out, wls = sys.argv[1], sys.argv[2:]
FILL = ("k_linear", "k_affine", "k_banded")
def per_fill(wl, counter):
    fs = glob.glob(f"{out}/{wl}_{counter}/**/*counter_collection.csv", recursive=True)
    if not fs: return None, []
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if r["Counter_Name"] != counter or not any(f in k for f in FILL): continue
        acc.setdefault(r["Dispatch_Id"], [k, 0.0])[1] += float(r["Counter_Value"])
    names = sorted({v[0] for v in acc.values()})
    if not names: return None, []
    # a batch may need more than one fill kernel per fill, each launched once per fill: fills = dispatches of any one of them
    fills = max(sum(1 for v in acc.values() if v[0] == nm) for nm in names)
    return sum(v[1] for v in acc.values()) / fills, names
traffic = {}
for wl in wls:
    w, names = per_fill(wl, "WRITE_SIZE")
    f, _ = per_fill(wl, "FETCH_SIZE")
    if w is None or f is None: continue
    res = {"workload": wl, "fill_kernels": names, "WRITE_SIZE_KB_per_fill": w, "FETCH_SIZE_KB_per_fill": f,
           "hbm_bytes_per_fill": w * 1024 + 2 * f * 1024,
           "note": "rocprofv3 --kernel-trace --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes over bench.py --steps 3 --warmup 1; "
                   "counters are KiB; FETCH_SIZE doubled (gfx950 reports half of wide streaming reads, MI355X_MICROARCH.md)"}
    try:
        b = json.load(open(f"{out}/{wl}_bench.json"))
        res["algorithmic_bytes_per_fill"] = b["roofline"]["algorithmic_bytes_per_launch"]
        res["traffic_over_algorithmic"] = round(res["hbm_bytes_per_fill"] / res["algorithmic_bytes_per_fill"], 4)
    except Exception:
        pass
    json.dump(res, open(f"{out}/{wl}_pmc.json", "w"), indent=1)
    traffic[wl] = int(res["hbm_bytes_per_fill"])
    print(wl, res.get("traffic_over_algorithmic"), names)
json.dump({"kernel_source_sha256": kernel_source_hash(), "workloads": traffic}, open(f"{out}/pmc_traffic.json", "w"), indent=1)
PY
for WL in $WLS; do python3 -c "
import json,sys
d=json.load(open('$OUT/${WL}_bench.json')); r=d['roofline']
print('$WL', d['value'], 'GCUPS', r['kernel_ms'], 'ms frac', r['frac'])"; head -3 $OUT/${WL}_kernel_stats.csv | cut -c1-150; done
