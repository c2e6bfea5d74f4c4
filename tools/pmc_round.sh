#!/bin/bash
# tools/pmc_round.sh <round> [workloads...] -- SQ / LDS / TCC / GRBM counters of the fill kernel of every BASELINE workload
# (tools/pmc.sh groups, one rocprofv3 --pmc pass per group, program directly after `--`) -> gpurun_out/pmc_<round>/<wl>/summary.txt
# (copy to profiles/<round>/<wl>_pmc_sq_tcc_summary.txt)
set -u
ROUND=$1; shift
WLS=${*:-lsw_10k_1024 lnw_10k_1024 lsw_1k_512 anw_1k_1024 bsw_10k_4096_b128 lnw_100k_short lsw_100k_short anw_100k_short}
for WL in $WLS; do
  mkdir -p gpurun_out/pmc_$ROUND; PMC_NO_TRAFFIC=1 tools/pmc.sh gpurun_out/pmc_$ROUND/$WL python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$ROUND/$WL.log 2>&1
  echo "pmc $WL done"
done
