"""GPU parity (bit-exact): the HIP fill kernels, called through the C ABI, against the CPU oracle on the same
seeded inputs -- scores, traceback start cells and every cell of the exported H / I / D matrices."""
import os

import numpy as np
import pytest

import oracle_py as O
from dpx_gpu_genomics_project_amd.synth import from_strings, make_batch, make_ragged_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "one-wave-per-pair", "one-wave-per-pair+ramp-lines", "one-wave-per-pair+four-wave-workgroups"])
def kernel_path(request, monkeypatch):
    """Every test of this file runs four times: with the engine's own choice (small multi-stripe batches take the split
    kernel), with DPX_SPLIT=0 (the one-wave-per-pair kernels incl. their rolling multi-stripe schedule), with the
    line-rounded ramp stores big batches use (DPX_RAMP_LINES=1) forced onto these small ones, and with the four-wave
    workgroups of launches of more than 4096 waves (DPX_WPB=4; small launches use one-wave workgroups)."""
    if request.param != "default":
        monkeypatch.setenv("DPX_SPLIT", "0")
    if request.param.endswith("ramp-lines"):
        monkeypatch.setenv("DPX_RAMP_LINES", "1")
    if request.param.endswith("four-wave-workgroups"):
        monkeypatch.setenv("DPX_WPB", "4")
    return request.param


HAND = [("0", "0"), ("0", "1"), ("0123", "0"), ("3", "0123"), ("00000000", "00000000"), ("01230123", "32103210"),
        ("ABxxxCDE", "ABCDE"), ("GTCATGCAATAACG", "ATGCAATA"), ("GTCAGTA", "ATACA"), ("4444", "4444"), ("0404", "4040"),
        ("", "0123"), ("0123", ""), ("", "")]


def _oracle(algo, refs, qry, w):
    if algo == "LSW":
        return O.lsw(refs, qry, w[0], w[1], w[2], want_dir=False)
    if algo == "LNW":
        return O.lnw(refs, qry, w[0], w[1], w[2], want_dir=False)
    return O.anw(refs, qry, w[0], w[1], w[2], w[3], want_dir=False)


def _check_batch(dpx, algo, sb, w, check_matrix_every=1):
    code = {"LNW": dpx.ALGO_LNW, "LSW": dpx.ALGO_LSW, "ANW": dpx.ALGO_ANW}[algo]
    ext = w[3] if len(w) > 3 else -1
    with dpx.Batch(code, sb.sequences, sb.pairs, w[0], w[1], w[2], ext) as b:
        b.fill()
        sc, er, ec = b.results()
        for p in range(sb.num_pairs):
            refs, qry = sb.ref(p), sb.qry(p)
            o = _oracle(algo, refs, qry, w)
            assert sc[p] == o.score, (algo, p, len(qry), len(refs))
            if algo == "LSW":
                assert (er[p], ec[p]) == (o.end_row, o.end_col), (algo, p)
            else:
                assert (er[p], ec[p]) == (len(qry), len(refs))
            if p % check_matrix_every == 0:
                H = b.matrix(p, dpx.MAT_H)
                assert np.array_equal(H.astype(np.int32), o.H), (algo, p, "H")
                if algo == "ANW":
                    assert np.array_equal(b.matrix(p, dpx.MAT_I).astype(np.int32), o.I), (algo, p, "I")
                    assert np.array_equal(b.matrix(p, dpx.MAT_D).astype(np.int32), o.D), (algo, p, "D")


@pytest.mark.parametrize("algo,w", [("LSW", (3, -1, -2)), ("LNW", (3, -1, -2)), ("ANW", (3, -1, -3, -1)),
                                    ("LSW", (5, -2, -3)), ("LNW", (5, -2, -3)), ("ANW", (2, -2, 0, -1))])
def test_hand_cases_including_empty(gpu, algo, w):
    _check_batch(gpu, algo, from_strings(HAND), w)


@pytest.mark.parametrize("R", ["2", "4", "8", "16"])
@pytest.mark.parametrize("algo,w", [("LSW", (3, -1, -2)), ("LNW", (3, -1, -2)), ("ANW", (3, -1, -3, -1))])
def test_tile_heights_and_stripe_edges(gpu, algo, w, R, monkeypatch):
    """Every rows-per-lane variant, on shapes that straddle the 64*R stripe edge and the 64-lane skew."""
    if algo == "ANW" and R == "16":
        pytest.skip("affine kernel is built for R <= 8")
    monkeypatch.setenv("DPX_R", R)
    r = int(R)
    shapes = [(1, 1), (3, 70), (63, 64), (64, 63), (65, 130), (64 * r, 100), (64 * r + 1, 67), (2 * 64 * r + 5, 90), (130, 5)]
    for i, (m, n) in enumerate(shapes):
        _check_batch(gpu, algo, make_batch(3, m, n, seed=100 + i), w)


@pytest.mark.parametrize("R", ["2", "4", "8"])
@pytest.mark.parametrize("algo,w", [("LSW", (3, -1, -2)), ("LNW", (3, -1, -2)), ("ANW", (3, -1, -3, -1))])
def test_rolling_schedule_multi_stripe(gpu, algo, w, R, monkeypatch):
    """References >= 128 columns with 2-4 stripes take the rolling schedule (lanes run on into the next stripe):
    full and partial last stripes, exact multiples, and a ragged mix that shares one group-interleaved block."""
    monkeypatch.setenv("DPX_R", R)
    r = int(R)
    for i, (m, n) in enumerate([(2 * 64 * r, 128), (3 * 64 * r - 7, 150), (2 * 64 * r + 1, 129), (4 * 64 * r, 131)]):
        if m * n > 600000:
            continue
        _check_batch(gpu, algo, make_batch(3, m, n, seed=200 + i, first_index=95), w)
    _check_batch(gpu, algo, make_ragged_batch(40, 64 * r + 1, 3 * 64 * r, 128, 200, seed=17), w, check_matrix_every=5)


@pytest.mark.parametrize("algo,w", [("LSW", (3, -1, -2)), ("LNW", (3, -1, -2)), ("ANW", (3, -1, -3, -1))])
def test_ragged_short_reads(gpu, algo, w):
    """cfg1-like ragged batch (reference 100-160, query 80-130) -> exercises the longest-first launch order."""
    _check_batch(gpu, algo, make_ragged_batch(200, 80, 130, 100, 160, seed=6), w, check_matrix_every=7)


def test_lsw_512_config(gpu):
    """BASELINE.json configs[1] shape (512x512 LSW), a 24-pair slice incl. the random (97th) and identical (101st) pairs."""
    sb = make_batch(24, 512, 512, seed=2, first_index=90)
    _check_batch(gpu, "LSW", sb, (3, -1, -2), check_matrix_every=4)


def test_lsw_1024_headline_shape(gpu):
    sb = make_batch(8, 1024, 1024, seed=1, first_index=95)
    _check_batch(gpu, "LSW", sb, (3, -1, -2), check_matrix_every=3)


def test_anw_1024_config(gpu):
    sb = make_batch(4, 1024, 1024, seed=3, first_index=95)
    _check_batch(gpu, "ANW", sb, (3, -1, -3, -1), check_matrix_every=2)


def test_lnw_1024(gpu):
    sb = make_batch(4, 1024, 1024, seed=5, first_index=95)
    _check_batch(gpu, "LNW", sb, (3, -1, -2), check_matrix_every=2)


def test_score_only_matches(gpu):
    sb = make_batch(16, 300, 260, seed=9)
    for algo, code, w in [("LSW", gpu.ALGO_LSW, (3, -1, -2, -1)), ("LNW", gpu.ALGO_LNW, (3, -1, -2, -1)), ("ANW", gpu.ALGO_ANW, (3, -1, -3, -1))]:
        with gpu.Batch(code, sb.sequences, sb.pairs, *w, flags=gpu.SCORE_ONLY) as b:
            b.fill()
            sc, er, ec = b.results()
            with pytest.raises(gpu.DpxError):
                b.matrix(0)
        for p in range(sb.num_pairs):
            assert sc[p] == _oracle(algo, sb.ref(p), sb.qry(p), w).score


def test_int16_range_is_enforced(gpu):
    sb = make_batch(1, 64, 64, seed=1)
    with pytest.raises(gpu.DpxError) as e:
        gpu.Batch(gpu.ALGO_LSW, sb.sequences, sb.pairs, 1000, -1, -2)
    assert e.value.status == -4


@pytest.mark.parametrize("algo,w", [("LSW", (3, -1, -2)), ("ANW", (3, -1, -3, -1))])
def test_long_reads_4096_need_more_than_64k_lds(gpu, algo, w):
    """4096x4096 pairs: the workgroup's dynamic LDS request exceeds the 64 KiB default (hipFuncSetAttribute path),
    8 stripes roll into each other, and scores approach the int16 ceiling (identical pair: 12288)."""
    sb = make_batch(2, 4096, 4096, seed=41, first_index=100)   # pair 0 identical, pair 1 mutated
    _check_batch(gpu, algo, sb, w)


def test_sequences_too_long_for_lds_are_refused(gpu):
    sb = make_batch(1, 10000, 10000, seed=1)
    with pytest.raises(gpu.DpxError) as e:
        gpu.Batch(gpu.ALGO_LSW, sb.sequences, sb.pairs, 3, -1, -2)
    assert e.value.status == -8   # DPX_ERR_UNSUPPORTED, never a silent fallback
