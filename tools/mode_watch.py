#!/usr/bin/env python3
"""tools/mode_watch.py [workload] [windows] [trials] -- does the fast / slow mode of a fill belong to the allocation or to the moment?
One process: a batch is created, filled in windows of 30 fills (time per window printed), destroyed, re-created ... If the time
flips between windows of ONE allocation, the mode is not a property of the pool.  Development aid (round 3)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "anw_1k_1024"
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 40
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 4
shapes = {"lsw_10k_1024": (dpx.ALGO_LSW, 10000, 1024, 1024, (3, -1, -2)), "anw_1k_1024": (dpx.ALGO_ANW, 1000, 1024, 1024, (3, -1, -3, -1)),
          "lsw_1k_512": (dpx.ALGO_LSW, 1000, 512, 512, (3, -1, -2))}
algo, npairs, m, n, w = shapes[wl]
dpx.init(0)
sb = dpx.make_batch(npairs, m, n, seed=1)
lib = dpx.load()
for trial in range(trials):
    with dpx.Batch(algo, sb.sequences, sb.pairs, *w) as b:
        ts = []
        for k in range(windows):
            ts.append(b.fill_timed(30))
            if k == windows // 2:
                time.sleep(0.5)  # an idle gap in the middle: does the mode change across it?
        print(f"{wl} trial {trial}: " + " ".join(f"{t:.0f}" for t in ts), flush=True)
    lib.dpx_shutdown()
    dpx.init(0)
