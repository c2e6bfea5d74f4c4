"""Short-read batch (the reference's own dataset shape: ~100k pairs, reference 100-160, query 80-130): fill + traceback
timing through the C ABI (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
dpx.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
t = time.time(); sb = make_ragged_batch(N, 80, 130, 100, 160, seed=6); print(f"synth {time.time()-t:.1f}s cells {sb.cells/1e9:.3f}e9", flush=True)
for algo in (dpx.ALGO_LNW, dpx.ALGO_LSW, dpx.ALGO_ANW):
    for flags in (0, 1):
        b = dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, -3 if algo == dpx.ALGO_ANW else -2, -1, flags=flags)
        b.fill_timed(1)
        us = min(b.fill_timed(3) for _ in range(3))
        extra = ""
        if not flags:
            t = time.time(); b.traceback(0); t1 = time.time() - t
            b.fill(); b.sync(); t = time.time(); b.traceback(0); t2 = time.time() - t
            extra = f"  traceback(all pairs)+D2H first {1e3*t1:.1f} ms, again {1e3*t2:.1f} ms"
        info = b.info()
        print(f"{dpx.ALGO_NAMES[algo]} flags={flags}: {us/1e3:.3f} ms  {sb.cells/us/1e3:.1f} GCUPS  alg {info['algorithmic_bytes']/us/1e3:.0f} GB/s  mat {info['matrix_bytes']/1e9:.2f} GB{extra}", flush=True)
        b.close()
