#!/usr/bin/env python3
"""tools/lanes_threshold.py -- where do the lane-packed kernels beat the engine's other choices?  Ragged short reads at several batch
sizes and a few mid-size shapes, DPX_LANES=0 / 1 (development aid; sets the thresholds in dpx_capi.cpp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
dpx.init(0)
cases = [(f"short x{n}", make_ragged_batch(n, 80, 130, 100, 160, seed=6)) for n in (512, 1024, 2048, 4096, 8192, 16384)]
cases += [("20000 x 250x300", dpx.make_batch(20000, 250, 300, seed=1)), ("20000 x 300x300", dpx.make_batch(20000, 300, 300, seed=1)),
          ("8000 x 500x500", dpx.make_batch(8000, 500, 500, seed=1)), ("20000 x 180x200", dpx.make_batch(20000, 180, 200, seed=1))]
for name, sb in cases:
    out = []
    for lanes in ("0", "1"):
        os.environ["DPX_LANES"] = lanes
        with dpx.Batch(dpx.ALGO_LSW, sb.sequences, sb.pairs, 3, -1, -2) as b:
            k = b.describe()["kernel"]
            b.fill_timed(3)
            t = min(b.fill_timed(5) for _ in range(3))
        out.append(f"{k} {t:8.1f} us")
    print(f"{name:18s}  " + "   |   ".join(out), flush=True)
