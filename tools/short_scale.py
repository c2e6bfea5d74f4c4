#!/usr/bin/env python3
"""tools/short_scale.py -- short-read fills (ref 100-160 x query 80-130) at growing batch sizes: what a launch's ramp-up and
tail cost the 100k batch (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx  # noqa: E402
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch  # noqa: E402

dpx.init(0)
for name in ("LNW", "LSW"):
    algo = {"LSW": dpx.ALGO_LSW, "LNW": dpx.ALGO_LNW}[name]
    for count in ([int(x) for x in sys.argv[1:]] or [25000, 50000, 100000, 200000, 400000]):
        sb = make_ragged_batch(count, 80, 130, 100, 160, seed=6)
        with dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, -2) as b:
            d = b.describe()
            b.fill_timed(10)
            t = min(b.fill_timed(20) for _ in range(3))
        print(f"{name} {count:7d} short pairs  {d['kernel']}  {t:9.1f} us  {sb.cells / t / 1e3:7.0f} GCUPS", flush=True)
