"""Quick single-GPU fill timing (development aid; bench.py is the contract)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dpx_gpu_genomics_project_amd as dpx

def run(algo, npairs, m, n, reps=3, flags=0, ext=-1, gap=-2):
    sb = dpx.make_batch(npairs, m, n, seed=1)
    t0 = time.time()
    b = dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, gap, ext, flags=flags)
    b.fill_timed(1)
    us = b.fill_timed(reps)
    info = b.info()
    gcups = info["cells"] / us / 1e3
    gbs = info["algorithmic_bytes"] / us / 1e3
    print(f"{dpx.ALGO_NAMES[algo]} R={os.environ.get('DPX_R','auto')} pairs={npairs} {m}x{n} flags={flags}: {us/1e3:.3f} ms  {gcups:.1f} GCUPS  {gbs:.1f} GB/s alg  (setup {time.time()-t0:.1f}s)", flush=True)
    b.close()

if __name__ == "__main__":
    dpx.init(0)
    print(dpx.device_info(), flush=True)
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    for R in (os.environ.get("DPX_RS", "8,16,4")).split(","):
        os.environ["DPX_R"] = R
        run(dpx.ALGO_LSW, npairs, 1024, 1024)
        run(dpx.ALGO_LSW, npairs, 1024, 1024, flags=dpx.SCORE_ONLY)
    os.environ["DPX_R"] = "8"
    run(dpx.ALGO_LNW, npairs, 1024, 1024)
    run(dpx.ALGO_ANW, max(npairs // 4, 1), 1024, 1024, ext=-1, gap=-3)
    if len(sys.argv) > 2:
        nb = int(sys.argv[2])
        sb = dpx.make_batch(nb, 4096, 4096, seed=4)
        b = dpx.Batch(dpx.ALGO_BSW, sb.sequences, sb.pairs, 3, -1, -2, band=128)
        b.fill_timed(1); us = b.fill_timed(3); info = b.info()
        print(f"BSW band128 pairs={nb} 4096x4096: {us/1e3:.3f} ms  {info['cells']/us/1e3:.1f} GCUPS(full-matrix cells)  in-band {nb*1028224/us/1e3:.1f} GCUPS  {info['algorithmic_bytes']/us/1e3:.1f} GB/s alg", flush=True)
        b.close()
