"""Device time of the traceback kernels alone (development aid): fill + output_begin/_end of a few batch shapes; run it under
`rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 tools/tb_kernel.py` (tools/tb_kernel.sh reads the trace).
TB_SHAPES="pairs:m:n,..." picks the shapes ("short" = 100k ragged short reads); DPX_TB_WALK=0|1|2 forces a walk."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
dpx.init(0)
spec = os.environ.get("TB_SHAPES", "1712:1024:1024,5000:1024:1024,4000:512:512")
for item in spec.split(","):
    if item == "short":
        name, sb = "100k short", make_ragged_batch(100000, 80, 130, 100, 160, seed=6)
    else:
        npairs, m, n = (int(x) for x in item.split(":"))
        name, sb = f"{npairs} x {m}x{n}", dpx.make_batch(npairs, m, n, seed=1)
    for algo in [{"LSW": dpx.ALGO_LSW, "LNW": dpx.ALGO_LNW, "ANW": dpx.ALGO_ANW}[x] for x in os.environ.get("TB_ALGOS", "LSW,LNW,ANW").split(",")]:
        b = dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, -3 if algo == dpx.ALGO_ANW else -2, -1)
        ts = []
        for _ in range(3):
            b.fill(); b.sync()
            t = time.perf_counter(); b.output_begin(0); b.output_end(); ts.append(time.perf_counter() - t)
        print(f"{name} {dpx.ALGO_NAMES[algo]}: output (traceback + text + D2H) {1e3*min(ts):.3f} ms", flush=True)
        b.close()
