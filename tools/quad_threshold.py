"""Where does four-pairs-per-wave start to beat one-pair-per-wave on short reads?  (development aid; DPX_LANES=0/1 per run)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
dpx.init(0)
for N in (1024, 2048, 4096, 8192, 16384):
    sb = make_ragged_batch(N, 80, 130, 100, 160, seed=6)
    row = [f"N={N:6d}"]
    for algo in (dpx.ALGO_LNW, dpx.ALGO_ANW):
        for flags in (0, 1):
            for quad in ("0", "1"):
                os.environ["DPX_LANES"] = quad
                b = dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, -3 if algo == dpx.ALGO_ANW else -2, -1, flags=flags)
                b.fill_timed(2)
                us = min(b.fill_timed(5) for _ in range(3))
                b.close()
                row.append(f"{dpx.ALGO_NAMES[algo]} f{flags} q{quad} {us:7.1f}us")
    print("  ".join(row), flush=True)
