set -u
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh gpurun_out/pmc_lnw_short python3 bench.py --workload lnw_100k_short --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_lnw_short.log 2>&1
cat gpurun_out/pmc_lnw_short/summary.txt
