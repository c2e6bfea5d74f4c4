"""A fill on a CALLER's stream (dpx_batch_fill(b, stream): bench.py fills on torch's current stream).  Round 4: small batches upload
their inputs with one asynchronous copy on the batch's own stream, so a fill elsewhere has to be ordered behind it, and the
traceback / text kernels (the batch's stream again) behind that fill."""
import ctypes as C

import pytest

import oracle_py as O
from dpx_gpu_genomics_project_amd.synth import make_batch, make_ragged_batch

pytestmark = pytest.mark.gpu
W = (3, -1, -2)


@pytest.mark.parametrize("algo", ["LSW", "LNW"])
def test_small_batch_filled_on_a_foreign_stream_right_after_create(gpu, algo):
    hip = C.CDLL("libamdhip64.so")                                 # the runtime the engine itself is linked against
    handle = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(handle), 1) == 0 and handle.value   # hipStreamNonBlocking
    code = {"LSW": gpu.ALGO_LSW, "LNW": gpu.ALGO_LNW}[algo]
    for rep in range(20):                                          # (a race would not show every time)
        sb = make_ragged_batch(7, 20, 300, 30, 260, seed=900 + rep) if rep % 2 else make_batch(5, 200, 180, seed=900 + rep, first_index=97)
        with gpu.Batch(code, sb.sequences, sb.pairs, *W) as b:
            b.fill(handle.value)                                   # no synchronisation between create and this fill
            sc, er, ec = b.results()
            for p in range(sb.num_pairs):
                o = (O.lsw if algo == "LSW" else O.lnw)(sb.ref(p), sb.qry(p), *W)
                end = (o.end_row, o.end_col) if algo == "LSW" else (len(sb.qry(p)), len(sb.ref(p)))
                assert (sc[p], er[p], ec[p]) == (o.score, *end), (rep, p)
                want = (("", "", "") if o.score == 0 else O.lsw_traceback(sb.ref(p), sb.qry(p), o)) if algo == "LSW" else O.lnw_traceback(sb.ref(p), sb.qry(p), o)
                assert b.traceback(p) == want, (rep, p)
    assert hip.hipStreamDestroy(handle) == 0
