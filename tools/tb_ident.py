import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_batch
dpx.init(0)
sb = make_batch(1712, 1024, 1024, seed=1)
# variant: queries identical to the references (pure diagonal paths)
sb2 = make_batch(1712, 1024, 1024, seed=1)
seq = sb2.sequences
for p in range(sb2.num_pairs):
    r = sb2.pairs[p]
    seq[r["queryIdx"]:r["queryIdx"] + 1024] = seq[r["referenceIdx"]:r["referenceIdx"] + 1024]
for name, b_ in (("mutated", sb), ("identical", sb2)):
    b = dpx.Batch(dpx.ALGO_LSW, b_.sequences, b_.pairs, 3, -1, -2)
    for _ in range(3):
        b.fill(); b.sync(); b.output_begin(0); b.output_end()
    print(name, "done", flush=True)
    b.close()
