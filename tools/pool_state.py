#!/usr/bin/env python3
"""tools/pool_state.py -- does the fast / slow state of the headline fill (profiles/README.md, run-to-run spread) follow the
address of the matrix pool?  One process: fill timings of the same batch with the pool freed and re-allocated behind
dummy allocations of different sizes (DPX_TRACE=1 prints the pool address)."""
import ctypes as C
import os
import sys

os.environ["DPX_TRACE"] = "1"
os.environ.setdefault("DPX_POOL_PROBE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx  # noqa: E402

hip = C.CDLL("libamdhip64.so")
dpx.init(0)
sb = dpx.make_batch(10000, 1024, 1024, seed=1)
lib = dpx.load()
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    dummy = C.c_void_p()
    size = (trial * 1234567 * 512) % (6 << 30)
    if size:
        assert hip.hipMalloc(C.byref(dummy), C.c_size_t(size)) == 0
    with dpx.Batch(dpx.ALGO_LSW, sb.sequences, sb.pairs, 3, -1, -2) as b:
        b.fill_timed(20)
        t = min(b.fill_timed(30) for _ in range(3))
    print(f"trial {trial}: dummy {size >> 20} MiB  fill {t:.1f} us  {sb.cells / t / 1e3:.1f} GCUPS", flush=True)
    lib.dpx_shutdown()   # frees the parked pool: the next trial allocates afresh
    dpx.init(0)
    if size:
        hip.hipFree(dummy)
