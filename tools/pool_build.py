import os, sys, time
sys.path.insert(0, os.getcwd())
import dpx_gpu_genomics_project_amd as dpx
dpx.init(0)
lib = dpx.load()
for gb in (1, 4, 4, 12, 22):
    lib.dpx_shutdown(); dpx.init(0)
    t = time.perf_counter(); rc = lib.dpx_pool_reserve(gb << 30, 1); dt = time.perf_counter() - t
    t = time.perf_counter(); lib.dpx_shutdown(); dt2 = time.perf_counter() - t
    print(f"{gb} GiB: reserve {dt*1e3:.1f} ms (rc {rc}), teardown {dt2*1e3:.1f} ms", flush=True)
    dpx.init(0)
