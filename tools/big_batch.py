"""One very large short-read batch (1M pairs) through fill + traceback, spot-checked against the oracle (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch
import oracle_py as O
dpx.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sb = make_ragged_batch(N, 80, 130, 100, 160, seed=9)
for algo, name in ((dpx.ALGO_LSW, "LSW"), (dpx.ALGO_ANW, "ANW")):
    t = time.time()
    b = dpx.Batch(algo, sb.sequences, sb.pairs, 3, -1, -3 if name == "ANW" else -2, -1)
    us = b.fill_timed(1); us = b.fill_timed(2)
    sc, er, ec = b.results()
    rng = np.random.default_rng(1)
    for p in list(rng.integers(0, N, 40)) + [0, N - 1]:
        refs, qry = sb.ref(int(p)), sb.qry(int(p))
        if name == "LSW":
            o = O.lsw(refs, qry, 3, -1, -2); want = ("", "", "") if o.score == 0 else O.lsw_traceback(refs, qry, o)
            assert (sc[p], er[p], ec[p]) == (o.score, o.end_row, o.end_col)
        else:
            o = O.anw(refs, qry, 3, -1, -3, -1); want = O.anw_traceback(refs, qry, o)
            assert sc[p] == o.score
        assert b.traceback(int(p)) == want
        assert np.array_equal(b.matrix(int(p)).astype(np.int32), o.H)
    info = b.info()
    print(f"{name} {N} pairs: fill {us/1e3:.2f} ms, {sb.cells/us/1e3:.0f} GCUPS, matrices {info['matrix_bytes']/1e9:.1f} GB, total {time.time()-t:.1f} s, 42 pairs verified", flush=True)
    b.close()
