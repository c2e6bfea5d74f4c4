"""BASELINE.json's configurations at their FULL sizes, through properties that do not need a CPU pass over every cell:
closed-form scores of the identical pairs every batch contains, agreement between the kernels that can fill the same
pair (packed two-per-wave, one-per-wave, quad), independence of a pair's result from the batch around it, idempotence
of a refill, and the oracle on a sample (whole batches where the oracle finishes in seconds)."""
import numpy as np
import pytest

import oracle_py as O
from dpx_gpu_genomics_project_amd.synth import make_batch, make_ragged_batch

pytestmark = pytest.mark.gpu
W = (3, -1, -2)


def _ident(sb, first_index=0):
    """Pairs whose query is a copy of the reference (synth.make_batch: every 101st, unless it is also a 97th = random)."""
    return [p for p in range(sb.num_pairs) if (first_index + p) % 101 == 100 and (first_index + p) % 97 != 96]


def _sub(sb, idx):
    """The pairs `idx` of `sb` as their own batch (same sequence buffer, fewer records)."""
    return sb.sequences, sb.pairs[np.asarray(idx)]


def test_headline_lsw_10k_1024(gpu, monkeypatch):
    sb = make_batch(10000, 1024, 1024, seed=1)
    with gpu.Batch(gpu.ALGO_LSW, sb.sequences, sb.pairs, *W) as b:          # default: packed two-pairs-per-wave kernel
        b.fill()
        sc, er, ec = b.results()
        sc2, er2, ec2 = (x.copy() for x in (sc, er, ec))
        b.fill()                                                             # idempotence of a refill
        sc, er, ec = b.results()
        assert np.array_equal(sc, sc2) and np.array_equal(er, er2) and np.array_equal(ec, ec2)
        for p in _ident(sb):                                                 # query == reference: 3 * 1024 at the last cell
            assert (sc[p], er[p], ec[p]) == (3 * 1024, 1024, 1024)
            ref_line, rel, qry_line = b.traceback(p)
            assert ref_line == qry_line == sb.ref(p).decode("latin-1") and rel == "*" * 1024
        assert sc.max() == 3 * 1024 and sc.min() > 0
        for p in range(0, 10000, 10):                                        # 1000 pairs: score and start cell against the oracle
            o = O.lsw(sb.ref(p), sb.qry(p), *W, want_dir=False)
            assert (sc[p], er[p], ec[p]) == (o.score, o.end_row, o.end_col), p
        rng = np.random.default_rng(5)
        sample = sorted(set(int(x) for x in rng.integers(0, 10000, 12)) | {0, 9999, 96})
        for p in sample[:6]:                                                 # every cell of a few pairs against the oracle
            o = O.lsw(sb.ref(p), sb.qry(p), *W)
            assert (sc[p], er[p], ec[p]) == (o.score, o.end_row, o.end_col)
            assert np.array_equal(b.matrix(p).astype(np.int32), o.H)
            assert b.traceback(p) == (("", "", "") if o.score == 0 else O.lsw_traceback(sb.ref(p), sb.qry(p), o))
    # the same pairs alone in a small batch take the one-pair-per-wave int32 kernel: same answers, whatever surrounds them
    seqs, prs = _sub(sb, sample)
    with gpu.Batch(gpu.ALGO_LSW, seqs, prs, *W) as small:
        small.fill()
        s1, r1, c1 = small.results()
    assert np.array_equal(s1, sc[sample]) and np.array_equal(r1, er[sample]) and np.array_equal(c1, ec[sample])
    # the whole batch again on the one-pair-per-wave kernel: all 10k results agree with the packed kernel
    monkeypatch.setenv("DPX_PACKED", "0")
    with gpu.Batch(gpu.ALGO_LSW, sb.sequences, sb.pairs, *W) as b:
        b.fill()
        s0, r0, c0 = b.results()
    assert np.array_equal(s0, sc) and np.array_equal(r0, er) and np.array_equal(c0, ec)


def test_config1_lsw_1k_512_every_pair_against_the_oracle(gpu):
    sb = make_batch(1000, 512, 512, seed=2)
    with gpu.Batch(gpu.ALGO_LSW, sb.sequences, sb.pairs, *W) as b:
        b.fill()
        sc, er, ec = b.results()
        for p in range(1000):
            o = O.lsw(sb.ref(p), sb.qry(p), *W, want_dir=False)
            assert (sc[p], er[p], ec[p]) == (o.score, o.end_row, o.end_col), p
            if p % 100 == 0:
                assert np.array_equal(b.matrix(p).astype(np.int32), o.H)


def test_config2_anw_1k_1024(gpu):
    sb = make_batch(1000, 1024, 1024, seed=3)
    w = (3, -1, -3, -1)
    with gpu.Batch(gpu.ALGO_ANW, sb.sequences, sb.pairs, *w) as b:
        b.fill()
        sc, er, ec = b.results()
        assert np.all(er == 1024) and np.all(ec == 1024)
        for p in _ident(sb):
            assert sc[p] == 3 * 1024
        for p in range(0, 1000):                                             # every pair: score and printed lines; every 8th all three matrices (round 3: 125 scores, 5 x 3 matrices)
            o = O.anw(sb.ref(p), sb.qry(p), *w, want_dir=True)
            assert sc[p] == o.score, p
            assert b.traceback(p) == O.anw_traceback(sb.ref(p), sb.qry(p), o), p
            if p % 8 == 0:
                for which, want in ((gpu.MAT_H, o.H), (gpu.MAT_I, o.I), (gpu.MAT_D, o.D)):
                    assert np.array_equal(b.matrix(p, which).astype(np.int32), want), (p, which)


def test_config3_banded_10k_4096_band128(gpu):
    sb = make_batch(10000, 4096, 4096, seed=4)
    with gpu.Batch(gpu.ALGO_BSW, sb.sequences, sb.pairs, *W, band=128) as b:
        b.fill()
        sc, er, ec = b.results()
        for p in _ident(sb):                                                 # the main diagonal lies inside any band
            assert (sc[p], er[p], ec[p]) == (3 * 4096, 4096, 4096)
        assert sc.max() == 3 * 4096 and sc.min() >= 0
        assert np.all(np.abs(er.astype(np.int64) - ec) <= 127)               # an end cell is an in-band cell
        sample = [0, 97, 4242, 9999]
        for p in list(range(3, 10000, 20)) + sample:                         # 1,028,224 in-band cells per pair (round 3: 504 pairs, was 44)
            o = O.lsw(sb.ref(p), sb.qry(p), *W, band=128, want_dir=False)
            assert (sc[p], er[p], ec[p]) == (o.score, o.end_row, o.end_col), p
        for p in [0, 96, 97, 100, 4242, 9999] + list(range(50, 10000, 250)):    # printed lines of 46 pairs (the wave walk over the band layout; round 3: 1),
            o = O.lsw(sb.ref(p), sb.qry(p), *W, band=128)                       # every cell of 6 whole matrices (round 3: 1) -- a random pair, a copy, mutated ones
            assert b.traceback(p) == (("", "", "") if o.score == 0 else O.lsw_traceback(sb.ref(p), sb.qry(p), o)), p
            if p in (0, 96, 97, 100, 4242, 9999):
                assert np.array_equal(b.matrix(p).astype(np.int32), o.H), p
    # restricting the paths can only lower a local score: banded <= unbanded, pair by pair
    seqs, prs = _sub(sb, sample)
    with gpu.Batch(gpu.ALGO_LSW, seqs, prs, *W) as full:
        full.fill()
        s_unbanded, _, _ = full.results()
    assert np.all(sc[sample] <= s_unbanded)


@pytest.mark.parametrize("algo", ["LNW", "LSW", "ANW"])
def test_config0_short_reads_100k_every_score_against_the_oracle(gpu, algo):
    """The reference's own dataset shape at full batch size: the quad kernels (four pairs per wave) by default."""
    sb = make_ragged_batch(100000, 80, 130, 100, 160, seed=6)
    w = (3, -1, -3, -1) if algo == "ANW" else (3, -1, -2, -1)
    code = {"LNW": gpu.ALGO_LNW, "LSW": gpu.ALGO_LSW, "ANW": gpu.ALGO_ANW}[algo]
    with gpu.Batch(code, sb.sequences, sb.pairs, *w) as b:
        b.fill()
        sc, er, ec = b.results()
        for p in range(0, 100000):                                           # every pair, every algorithm (round 3; LSW / ANW were every 7th)
            refs, qry = sb.ref(p), sb.qry(p)
            if algo == "LNW":
                assert sc[p] == O.lnw(refs, qry, *w[:3], want_dir=False).score, p
            elif algo == "LSW":
                o = O.lsw(refs, qry, *w[:3], want_dir=False)
                assert (sc[p], er[p], ec[p]) == (o.score, o.end_row, o.end_col), p
            else:
                assert sc[p] == O.anw(refs, qry, *w, want_dir=False).score, p
