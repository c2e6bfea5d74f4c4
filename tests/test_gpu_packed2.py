"""GPU parity of the 2-bit packed input (dpx_pack2 + dpx_batch_create_packed2; SURVEY 8f3, the input side of the reference's
c++/parseInput.cpp:78-112): a batch whose sequences arrive as four bases per byte is the byte batch -- same scores, start cells,
matrices, traceback lines and printed text -- for every algorithm, for ragged and empty sequences, for batches cut out of the
middle of a file (first_pair > 0: the device expands from a 16-base boundary), and through the batched driver (-pack2)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_py as O
from dpx_gpu_genomics_project_amd.synth import from_strings, make_batch, make_ragged_batch, write_pairs_file

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = {"LSW": (3, -1, -2, -1), "LNW": (3, -1, -2, -1), "ANW": (3, -1, -3, -1), "BSW": (3, -1, -2, -1)}


def _code(dpx, algo):
    return {"LNW": dpx.ALGO_LNW, "LSW": dpx.ALGO_LSW, "ANW": dpx.ALGO_ANW, "BSW": dpx.ALGO_BSW}[algo]


def _same(dpx, algo, sb, first=0, count=None, band=0, every=1):
    pk, al = dpx.pack2(sb.sequences, sb.pairs)
    kw = dict(band=band, first_pair=first, num_pairs=count)
    with dpx.Batch(_code(dpx, algo), sb.sequences, sb.pairs, *W[algo], **kw) as a, \
         dpx.Batch(_code(dpx, algo), None, sb.pairs, *W[algo], packed2=(pk, al, sb.sequences.size), **kw) as b:
        assert a.describe()["seq_input"] == "bytes" and b.describe()["seq_input"] == "packed2"
        a.fill(); b.fill()
        ra, rb = a.results(), b.results()
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y), algo
        for p in range(0, a.num_pairs, every):
            assert np.array_equal(a.matrix(p), b.matrix(p)), (algo, p)
            if algo != "BSW":
                assert a.traceback(p) == b.traceback(p), (algo, p)
        if algo != "BSW":
            a.output_begin(first); b.output_begin(first)
            (ta, oa), (tb, ob) = a.output_end(), b.output_end()
            assert ta == tb and np.array_equal(oa, ob) and len(ta) > 0
    return ra


@pytest.mark.parametrize("algo", ["LSW", "LNW", "ANW"])
def test_packed2_batches_are_the_byte_batches(gpu, algo):
    sb = make_ragged_batch(70, 1, 300, 1, 400, seed=501)
    sc, er, ec = _same(gpu, algo, sb, every=3)
    for p in (0, 7, 69):  # and both are the oracle's
        o = {"LSW": O.lsw, "LNW": O.lnw}.get(algo, None)
        o = o(sb.ref(p), sb.qry(p), *W[algo][:3]) if o else O.anw(sb.ref(p), sb.qry(p), *W[algo])
        assert sc[p] == o.score
    _same(gpu, algo, make_batch(6, 1024, 1024, seed=502), every=2)          # the packed-int16 / striped kernels
    _same(gpu, algo, make_ragged_batch(2100, 80, 130, 100, 160, seed=503), every=97)  # the lane-packed kernels


def test_packed2_banded(gpu):
    _same(gpu, "BSW", make_batch(5, 700, 700, seed=504), band=64)


def test_packed2_middle_of_a_file_and_empty_sequences(gpu):
    """first_pair > 0: the upload starts at the 16-base boundary below the batch's first base; empty sequences; ACGT."""
    sb = make_ragged_batch(90, 3, 70, 3, 90, seed=505)
    for first, count in ((1, 5), (17, 40), (89, 1)):
        _same(gpu, "LSW", sb, first=first, count=count)
        _same(gpu, "LNW", sb, first=first, count=count)
    acgt = from_strings([("GATTACA", "GCATGCT"), ("", "ACGT"), ("ACGT", ""), ("", ""), ("ACGTACGTTGCA" * 9, "TTGACGTACGAACGT" * 5), ("A", "A")])
    for algo in ("LSW", "LNW", "ANW"):
        _same(gpu, algo, acgt)


def test_pack2_refuses_a_fifth_symbol_and_bad_arguments(gpu):
    five = from_strings([("ACGTN", "ACGT")])
    with pytest.raises(gpu.DpxError) as e:
        gpu.pack2(five.sequences, five.pairs)
    assert e.value.status == -8
    sb = make_batch(2, 20, 20, seed=1)
    pk, al = gpu.pack2(sb.sequences, sb.pairs)
    with pytest.raises(gpu.DpxError):  # index + size beyond the number of bases
        gpu.Batch(gpu.ALGO_LSW, None, sb.pairs, 3, -1, -2, packed2=(pk, al, 10))


def test_batched_driver_with_pack2_prints_the_same_text(gpu, tmp_path):
    sb = make_ragged_batch(300, 40, 120, 50, 150, seed=506)
    f = tmp_path / "pairs.txt"
    write_pairs_file(sb, str(f))
    exe = os.path.join(ROOT, "dpx_gpu_genomics_project_amd", "hostcpp", "dpx_main")
    subprocess.run(["make", "-s", "-C", os.path.dirname(exe)], check=True)
    outs = []
    for extra in ([], ["-pack2"]):
        r = subprocess.run([exe, "-pairs", str(f), "-algo", "LNW", "-batch", "128"] + extra, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode()
        text = r.stdout.decode("latin-1")
        outs.append(text[text.index("Pair # | Score"):text.index("Elapsed time")])
    assert outs[0] == outs[1] and outs[0].count("\n") > 4 * 300
