#!/usr/bin/env python3
"""tools/pool_leak.py -- allocate and drop the chunked matrix pool many times; device memory in use must come back every time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx  # noqa: E402

hip = C.CDLL("libamdhip64.so")
dpx.init(0)
lib = dpx.load()


def free_gb():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 1e9


base = free_gb()
for k in range(12):
    assert lib.dpx_pool_reserve(C.c_size_t((3 + k) << 30), 2) == 0
    held = free_gb()
    lib.dpx_shutdown()
    dpx.init(0)
    print(f"round {k}: two pools of {3 + k} GiB: free {held:.1f} GB while parked, {free_gb():.1f} GB after shutdown (start {base:.1f})", flush=True)
assert free_gb() > base - 1.0, "device memory leaked"
print("no leak")
