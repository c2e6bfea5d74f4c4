mkdir -p gpurun_out/r3a
export DPX_POOL_CHUNK_MB=256
for P in 6144 8192 10000 12288 16384; do
  DPX_AB_PAIRS=$P DPX_AB_VARIANTS="DPX_GROUP=100000;DPX_GROUP=64" timeout -k 10 200 python3 tools/pool_ab.py 2 lsw_10k_1024
done
for WL in lnw_10k_1024 anw_1k_1024 bsw_10k_4096_b128 lsw_1k_512; do
  DPX_AB_VARIANTS="DPX_GROUP=64;DPX_GROUP=100000;DPX_GROUP=512" timeout -k 10 300 python3 tools/pool_ab.py 3 $WL
done
