#!/bin/bash
# tools/r3e_exp.sh -- round-3 experiments (development aid): batch-size scaling of the short-read fills, one wave per workgroup,
# non-temporal stores in the headline kernel, end-to-end traceback times
set -e
O=gpurun_out/r3e; mkdir -p $O
python3 tools/short_scale.py > $O/short_scale.txt 2>&1
DPX_LIB=$PWD/tools/bin/libdpxalign_wg64.so python3 tools/short_scale.py > $O/short_scale_wg64.txt 2>&1
for i in 1 2; do
  for L in dpx_gpu_genomics_project_amd/libdpxalign.so tools/bin/libdpxalign_nt.so tools/bin/libdpxalign_wg64.so; do
    for WL in lsw_10k_1024 lnw_100k_short; do
      printf "%s %s " $WL $L
      DPX_LIB=$PWD/$L python3 bench.py --workload $WL --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel_ms'], r['frac'])"
    done
  done
done > $O/ab_nt_wg64.txt 2>&1
tools/e2e.sh 10000 long > $O/e2e_long.txt 2>&1
tools/e2e.sh 10000 long >> $O/e2e_long.txt 2>&1
python3 tools/tb_time.py long > $O/tb_time_long.txt 2>&1
