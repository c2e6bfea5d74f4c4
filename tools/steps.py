"""Per-fill kernel time over many back-to-back fills (development aid): shows the DVFS ramp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx
dpx.init(0)
sb = dpx.make_batch(10000, 1024, 1024, seed=1)
b = dpx.Batch(dpx.ALGO_LSW, sb.sequences, sb.pairs, 3, -1, -2)
ts = [b.fill_timed(1) / 1e3 for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120)]
for i in range(0, len(ts), 10):
    print(i, " ".join(f"{t:.2f}" for t in ts[i:i + 10]), flush=True)
b.close()
