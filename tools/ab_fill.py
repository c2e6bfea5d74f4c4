#!/usr/bin/env python3
"""tools/ab_fill.py <workload> [pairs] -- mean fill time of one bench.py workload through the ROUND-1 subset of the C ABI
(create / fill_timed only), so that DPX_LIB can point at any older build of the library for A/B runs on one box."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import BAND, WORKLOADS  # noqa: E402
from dpx_gpu_genomics_project_amd.capi import PAIR_DTYPE, Params, lib_path  # noqa: E402
from dpx_gpu_genomics_project_amd.synth import make_batch, make_ragged_batch  # noqa: E402

wl = sys.argv[1]
algo_name, npairs, m, n, match, mismatch, go, ge, seed = WORKLOADS[wl]
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else abs(npairs)
sb = make_ragged_batch(npairs, 80, 130, 100, 160, seed=seed) if m == 0 else make_batch(npairs, m, n, seed=seed)
lib = C.CDLL(lib_path())
vp = C.c_void_p
lib.dpx_batch_create.argtypes = [C.POINTER(Params), vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(vp)]
lib.dpx_batch_fill_timed.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
lib.dpx_batch_destroy.argtypes = [vp]
assert lib.dpx_init(0) == 0
prm = Params({"LNW": 0, "LSW": 1, "ANW": 2, "BSW": 3}[algo_name], match, mismatch, go, ge, BAND if algo_name == "BSW" else 0)
h = vp()
seq = np.ascontiguousarray(sb.sequences); prs = np.ascontiguousarray(sb.pairs, dtype=PAIR_DTYPE)
assert lib.dpx_batch_create(C.byref(prm), seq.ctypes.data, seq.size, prs.ctypes.data, 0, npairs, 4, C.byref(h)) == 0  # DPX_TUNE_PLACEMENT (ignored by older builds)
us = C.c_double()
lib.dpx_batch_fill_timed(h, 20, C.byref(us))
best = []
for _ in range(3):
    lib.dpx_batch_fill_timed(h, 30, C.byref(us))
    best.append(us.value)
print(f"{os.path.basename(lib_path()):18s} {wl:20s} {npairs:7d} pairs  {min(best):9.1f} us  {sb.cells / min(best) / 1e3:8.1f} GCUPS")
lib.dpx_batch_destroy(h)
