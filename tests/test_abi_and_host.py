"""CPU-side checks of the boundary: libdpxalign.so loads and exports every symbol include/dpx_align.h declares,
fails loudly without a GPU, and the host-side plumbing (input layout, sharding) behaves like the reference's."""
import os
import re

import numpy as np
import pytest

import dpx_gpu_genomics_project_amd as dpx
from dpx_gpu_genomics_project_amd import capi
from dpx_gpu_genomics_project_amd.shard import shard_range
from dpx_gpu_genomics_project_amd.synth import from_strings, make_batch, make_ragged_batch, parse_pairs_file, write_pairs_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dpx_align.h")).read()
    declared = set(re.findall(r"\b(dpx_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.ABI_SYMBOLS), declared ^ set(capi.ABI_SYMBOLS)
    lib = dpx.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dpx_abi_version() == 1
    assert lib.dpx_strerror(-2).decode().startswith("no usable HIP device")


def test_no_cpu_fallback_without_gpu():
    if dpx.device_count() > 0:
        pytest.skip("a GPU is present")
    sb = make_batch(1, 8, 8, seed=1)
    with pytest.raises(dpx.DpxError) as e:
        dpx.Batch(dpx.ALGO_LSW, sb.sequences, sb.pairs)
    assert e.value.status == -2  # DPX_ERR_NO_DEVICE: the product never computes on the CPU


def test_pairs_file_roundtrip_matches_parseinput_layout(tmp_path):
    sb = make_ragged_batch(37, 5, 40, 5, 50, seed=3)
    path = str(tmp_path / "pairs.txt")
    write_pairs_file(sb, path)
    raw = open(path, "rb").read()
    assert raw.count(b"\n") == 3 * 37 and b"\0" not in raw
    back = parse_pairs_file(path)
    assert np.array_equal(back.sequences, sb.sequences)
    assert np.array_equal(back.pairs, sb.pairs)
    # parseInput.cpp:78-112: referenceIdx is the byte after line 0's newline, sizes exclude the terminator
    for p in range(back.num_pairs):
        r = back.pairs[p]
        assert back.sequences[r["referenceIdx"] + r["referenceSize"]] == 0
        assert back.sequences[r["queryIdx"] + r["querySize"]] == 0
        assert r["queryIdx"] == r["referenceIdx"] + r["referenceSize"] + 1
    assert parse_pairs_file(path, cap=5).num_pairs == 5


def test_pairs_file_rejects_bad_line_count(tmp_path):
    path = str(tmp_path / "bad.txt")
    open(path, "wb").write(b"0\nACGT\n")
    with pytest.raises(ValueError):  # parseInput.cpp:38-41 exits(1) here
        parse_pairs_file(path)


def test_synth_batch_composition():
    sb = make_batch(202, 64, 64, seed=1)
    assert sb.cells == 202 * 64 * 64
    assert sb.ref(100) == sb.qry(100)          # every 101st pair identical
    assert set(sb.ref(0)) <= set(b"0123")
    a = make_batch(10, 32, 48, seed=5)
    b = make_batch(10, 32, 48, seed=5)
    assert np.array_equal(a.sequences, b.sequences)   # seeded, reproducible
    fs = from_strings([("ACGT", "AC"), ("", "A")])
    assert fs.ref(0) == b"ACGT" and fs.qry(1) == b"A" and fs.ref(1) == b""


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 9, 100000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d


def test_reorder_output_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("reorder_output", os.path.join(ROOT, "tools", "reorder_output.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = "Parsing input file: x\nPair # | Score\n2 | 0\n\n\n\n0 | 5\nAC\n**\nAC\n1 | -3\nA_\n* \nAG\nElapsed time (usec): 7\nCleaning up\n"
    want = "Parsing input file: x\nPair # | Score\n0 | 5\nAC\n**\nAC\n1 | -3\nA_\n* \nAG\n2 | 0\n\n\n\nElapsed time (usec): 7\nCleaning up\n"
    assert mod.reorder(text) == want
