import sys; sys.path.insert(0, ".")
import dpx_gpu_genomics_project_amd as dpx
dpx.init(0)
lib = dpx.load()
lib.dpx_pool_reserve(4 << 30, 2)
sb = dpx.make_batch(1712, 1024, 1024, seed=1)
b1 = dpx.Batch(dpx.ALGO_LSW, sb.sequences, sb.pairs, 3, -1, -2)
b2 = dpx.Batch(dpx.ALGO_LSW, sb.sequences, sb.pairs, 3, -1, -2)
print(b1.describe()["pool_bytes"], b2.describe()["pool_bytes"], b1.describe()["kernel"], b1.describe()["waves_per_workgroup"])
for r in range(4):
    for name, b in (("pool1", b1), ("pool2", b2)):
        b.fill_timed(1)
        us = [b.fill_timed(1) for _ in range(5)]
        print(name, " ".join(f"{u:.0f}" for u in us), "us")
b1.close(); b2.close()
