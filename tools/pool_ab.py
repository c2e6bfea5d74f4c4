#!/usr/bin/env python3
"""tools/pool_ab.py [rounds] [workload] -- ONE process, one box: the same resident batch re-created on matrix pools of different
construction (DPX_POOL / DPX_POOL_CHUNK_MB / DPX_GROUP ...), alternating, fill time + memset time of the pool per variant.
Variants are ';'-separated 'K=V K=V' strings in DPX_AB_VARIANTS.  Development aid (round 3: what the pool is built from matters)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = sys.argv[2] if len(sys.argv) > 2 else "lsw_10k_1024"
variants = [v.strip() for v in os.environ.get("DPX_AB_VARIANTS", "DPX_POOL=malloc;DPX_POOL=vmm").split(";") if v.strip()]
shapes = {"lsw_10k_1024": (dpx.ALGO_LSW, 10000, 1024, 1024, (3, -1, -2)), "lnw_10k_1024": (dpx.ALGO_LNW, 10000, 1024, 1024, (3, -1, -2)),
          "anw_1k_1024": (dpx.ALGO_ANW, 1000, 1024, 1024, (3, -1, -3, -1)), "lsw_1k_512": (dpx.ALGO_LSW, 1000, 512, 512, (3, -1, -2))}
shapes["bsw_10k_4096_b128"] = (dpx.ALGO_BSW, 10000, 4096, 4096, (3, -1, -2))
algo, npairs, m, n, w = shapes[wl]
npairs = int(os.environ.get("DPX_AB_PAIRS", npairs))
band = 128 if algo == dpx.ALGO_BSW else 0
dpx.init(0)
sb = dpx.make_batch(npairs, m, n, seed=1)
lib = dpx.load()
keys = sorted({kv.split("=")[0] for v in variants for kv in v.split()})
for r in range(rounds):
    for v in variants:
        for k in keys:
            os.environ.pop(k, None)
        for kv in v.split():
            k, val = kv.split("=", 1)
            os.environ[k] = val
        with dpx.Batch(algo, sb.sequences, sb.pairs, *w, band=band, flags=dpx.TUNE_PLACEMENT) as b:
            d = b.describe()
            b.fill_timed(20)
            t = min(b.fill_timed(30) for _ in range(3))
        print(f"{wl} x{npairs} round {r} [{v:44s}] fill {t:8.1f} us  {sb.cells / t / 1e3:7.1f} GCUPS  pool={d.get('pool')} chunk={d.get('pool_chunk_mb')} memset_ms={d.get('pool_memset_ms')}", flush=True)
        lib.dpx_shutdown()  # frees the parked pool: the next variant allocates afresh
        dpx.init(0)
