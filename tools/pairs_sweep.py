#!/usr/bin/env python3
"""tools/pairs_sweep.py <algo> <m> <n> <pairs...> -- fill rate of uniform batches of the given sizes under the current environment
(development aid; run it under DPX_PACKED=1, DPX_SPLIT=0 ... to compare the engine's kernel choices at one batch size)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx  # noqa: E402

name, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dpx.init(0)
algo = {"LSW": dpx.ALGO_LSW, "LNW": dpx.ALGO_LNW, "ANW": dpx.ALGO_ANW}[name]
for count in map(int, sys.argv[4:]):
    sb = dpx.make_batch(count, m, n, seed=3)
    with dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, -3 if name == "ANW" else -2, -1) as b:
        d = b.describe()
        b.fill_timed(10)
        t = min(b.fill_timed(20) for _ in range(3))
    print(f"{name} {count:6d} x {m}x{n}  {d['kernel']:18s} R={d['rows_per_lane']:2d}  {t:9.1f} us  {sb.cells / t / 1e3:7.0f} GCUPS", flush=True)
