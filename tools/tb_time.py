"""Device traceback timing (kernel + D2H of the line buffers) for long and short pairs (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
dpx.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
shapes = []
if which in ("both", "long"):
    shapes.append(("5000 x 1024x1024", dpx.make_batch(5000, 1024, 1024, seed=1)))
if which in ("p512",):
    shapes.append(("4000 x 512x512", dpx.make_batch(4000, 512, 512, seed=1)))
    shapes.append(("1000 x 512x512", dpx.make_batch(1000, 512, 512, seed=1)))
    shapes.append(("3000 x 700x700", dpx.make_batch(3000, 700, 700, seed=1)))
if which in ("both", "mid"):
    shapes.append(("20000 x 300x300", dpx.make_batch(20000, 300, 300, seed=1)))
if which in ("both", "short"):
    shapes.append(("100k short", make_ragged_batch(100000, 80, 130, 100, 160, seed=6)))
for name, sb in shapes:
    for algo in (dpx.ALGO_LSW, dpx.ALGO_LNW, dpx.ALGO_ANW):
        b = dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, -3 if algo == dpx.ALGO_ANW else -2, -1)
        b.fill(); b.sync(); b.traceback(0)
        ts = []
        for _ in range(3):
            b.fill(); b.sync(); t = time.perf_counter(); b.traceback(0); ts.append(time.perf_counter() - t)
        print(f"{name:18s} {dpx.ALGO_NAMES[algo]}: traceback of all pairs + D2H {1e3*min(ts):.2f} ms", flush=True)
        b.close()
