"""ANW 1000 x 1024^2 (BASELINE configs[2]) by group size of the interleaved layout, ON ONE ALLOCATION (round 4):
the fill of this batch runs 0.94 or 1.20 ms depending on the allocation of its pool; does the address pattern of the fill decide which?
For each of several fresh pools: the groups in turn, twice, every batch re-using the parked pool (same address range)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx
dpx.init(0)
sb = dpx.make_batch(1000, 1024, 1024, seed=3)
groups = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,1,2,4,8,16,32,128,256,1000").split(",")]
for pool in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    dpx.load().dpx_shutdown(); dpx.init(0)  # drop the parked pool: the next batch builds a fresh one
    rows = []
    for rnd in range(2):
        for g in groups:
            os.environ["DPX_GROUP"] = str(g)
            b = dpx.Batch(dpx.ALGO_ANW, sb.sequences, sb.pairs, 3, -1, -3, -1)
            b.fill_timed(2)
            us = min(b.fill_timed(5) for _ in range(3))
            rows.append((g, us, b.describe()["pool_ranges"] if "pool_ranges" in b.describe() else ""))
            b.close()
    print(f"pool {pool}: " + "  ".join(f"g{g}={us/1e3:.3f}" for g, us, _ in rows), flush=True)
