"""No fill kernel writes behind its batch's matrices (round 3).  DPX_POOL_GUARD=1 puts 4 MiB of pattern behind the matrices of every
batch and dpx_batch_sync() fails if a byte of it changed -- whatever happens to be mapped behind a pool on a given day, an overshoot
shows up here and not as a GPU memory access fault.  Every kernel family, shapes at the edges of their tiles (rows and columns that
are not multiples of 8 / 16 / 64, one-column and one-row matrices, ragged waves, odd pair counts beside the couples)."""
import pytest

from dpx_gpu_genomics_project_amd.synth import from_strings, make_batch, make_ragged_batch

pytestmark = pytest.mark.gpu

CASES = [
    ("one wave per pair, 2/4/8/16 rows per lane", {"DPX_SPLIT": "0", "DPX_PACKED": "0"},
     [(7, 100, 131), (5, 250, 77), (4, 500, 513), (3, 1024, 1000), (3, 1000, 1024), (2, 1, 1), (2, 1, 300), (2, 300, 1)]),
    ("rolling multi-stripe schedule", {"DPX_SPLIT": "0"}, [(2, 2100, 700), (2, 1100, 130), (1, 4096, 4096)]),
    ("split", {"DPX_SPLIT": "1"}, [(5, 600, 500), (3, 257, 129), (3, 1024, 1024), (2, 2000, 300)]),
    ("packed couples + an odd pair", {"DPX_PACKED": "1"}, [(7, 1024, 1024), (5, 513, 700), (9, 120, 131), (3, 300, 8)]),
    ("lane-packed", {"DPX_LANES": "1"}, [(41, 100, 131), (9, 512, 300), (7, 1024, 99), (33, 17, 9)]),
    ("lane-packed int32", {"DPX_LANES": "1", "DPX_LANES_PK": "0"}, [(41, 100, 131), (9, 512, 300), (7, 1024, 99)]),
]


@pytest.mark.parametrize("algo", ["LSW", "LNW", "ANW"])
def test_no_fill_writes_behind_its_matrices(gpu, algo, monkeypatch):
    monkeypatch.setenv("DPX_POOL_GUARD", "1")
    code = {"LNW": gpu.ALGO_LNW, "LSW": gpu.ALGO_LSW, "ANW": gpu.ALGO_ANW}[algo]
    w = (3, -1, -3, -1) if algo == "ANW" else (3, -1, -2, -1)
    for name, env, shapes in CASES:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for count, m, n in shapes:
            if algo == "ANW" and m > 512 and "lane" in name:
                continue
            with gpu.Batch(code, *(lambda sb: (sb.sequences, sb.pairs))(make_batch(count, m, n, seed=m + n)), *w) as b:
                b.fill()
                b.sync()  # raises DpxError if the guard band was touched
        with gpu.Batch(code, *(lambda sb: (sb.sequences, sb.pairs))(make_ragged_batch(300, 20, 300, 30, 400, seed=5)), *w) as b:
            b.fill(); b.sync()
        for k in env:
            monkeypatch.delenv(k)
    sb = make_ragged_batch(9000, 80, 130, 100, 160, seed=6)          # the default path of short reads (>= 2048 pairs)
    with gpu.Batch(code, sb.sequences, sb.pairs, *w) as b:
        b.fill(); b.sync()


def test_no_banded_fill_writes_behind_its_matrices(gpu, monkeypatch):
    monkeypatch.setenv("DPX_POOL_GUARD", "1")
    for count, m, n, band, env in [(5, 700, 700, 64, {}), (4, 4096, 4096, 128, {}), (6, 300, 900, 33, {}), (6, 700, 700, 64, {"DPX_PACKED": "1"})]:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sb = make_batch(count, m, n, seed=band)
        with gpu.Batch(gpu.ALGO_BSW, sb.sequences, sb.pairs, 3, -1, -2, band=band) as b:
            b.fill(); b.sync()
        for k in env:
            monkeypatch.delenv(k)


def test_the_guard_notices_an_overwrite(gpu, monkeypatch):
    """The checker itself: DPX_POOL_GUARD=selftest plants one wrong byte in the band."""
    monkeypatch.setenv("DPX_POOL_GUARD", "selftest")
    sb = make_batch(3, 200, 200, seed=1)
    with gpu.Batch(gpu.ALGO_LSW, sb.sequences, sb.pairs, 3, -1, -2) as b:
        b.fill()
        with pytest.raises(gpu.DpxError) as e:
            b.sync()
        assert "12345" in str(e.value) or e.value.status == -3
