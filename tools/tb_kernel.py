"""Device time of the traceback kernels alone (development aid): fill + output_begin/_end of a few batch shapes; run it under
`rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 tools/tb_kernel.py` and read the k_traceback* rows with
tools/timeline.py DIR, or give --print to see host-side times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx
dpx.init(0)
shapes = [(1712, 1024, 1024), (5000, 1024, 1024), (4000, 512, 512)]
for npairs, m, n in shapes:
    sb = dpx.make_batch(npairs, m, n, seed=1)
    for algo in (dpx.ALGO_LSW, dpx.ALGO_LNW, dpx.ALGO_ANW):
        b = dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, -3 if algo == dpx.ALGO_ANW else -2, -1)
        ts = []
        for _ in range(3):
            b.fill(); b.sync()
            t = time.perf_counter(); b.output_begin(0); b.output_end(); ts.append(time.perf_counter() - t)
        print(f"{npairs} x {m}x{n} {dpx.ALGO_NAMES[algo]}: output (traceback + text + D2H) {1e3*min(ts):.3f} ms", flush=True)
        b.close()
