"""How does the fill time evolve inside the FIRST GPU process on a fresh box?  (development aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx
dpx.init(0)
sb = dpx.make_batch(10000, 1024, 1024, seed=1)
b = dpx.Batch(dpx.ALGO_LSW, sb.sequences, sb.pairs, 3, -1, -2, -1)
t0 = time.time()
for k in range(14):
    us = b.fill_timed(20)
    print(f"t={time.time()-t0:6.2f}s  fills {20*k:4d}-{20*k+19:4d}: {us/1e3:.3f} ms/fill  {sb.cells/us/1e3:.0f} GCUPS", flush=True)
    if k == 6:
        time.sleep(3.0); print("  (slept 3 s)", flush=True)
b.close()
