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
    declared = set(re.findall(r"\b(dpx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.ABI_SYMBOLS), declared ^ set(capi.ABI_SYMBOLS)
    lib = dpx.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dpx_abi_version() == 3
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


def _fnv(sb, lo, hi):
    h = 1469598103934665603
    for p in range(lo, hi):
        for part in (sb.ref(p), sb.qry(p)):
            for c in part:
                h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return format(h, "x")


def test_cpp_loaders_agree_with_the_file_format(tmp_path):
    """hostcpp/parseInput.cpp (the reference's loader, mirrored) and parseInputShard (mmap, one pass, only the rank's
    pairs) against the Python restatement of the format, for the whole file and for every rank of 1-, 3- and 8-way splits."""
    import json
    import subprocess
    host = os.path.join(ROOT, "dpx_gpu_genomics_project_amd", "hostcpp")
    subprocess.run(["make", "-s", "-C", host, "parse_tool"], check=True)
    tool = os.path.join(host, "parse_tool")
    sb = make_ragged_batch(101, 1, 60, 1, 70, seed=8)
    path = str(tmp_path / "pairs.txt")
    write_pairs_file(sb, path)
    whole = json.loads(subprocess.run([tool, path], capture_output=True, text=True, check=True).stdout)
    sizes = [[int(r["referenceSize"]), int(r["querySize"])] for r in sb.pairs]
    assert whole["numPairs"] == whole["totalPairs"] == 101 and whole["sizes"] == sizes
    assert whole["numBytes"] == os.path.getsize(path) and whole["fnv"] == _fnv(sb, 0, 101)
    assert whole["numCells"] == sum(a * b for a, b in sizes)
    for world in (1, 3, 8):
        seen = 0
        for rank in range(world):
            part = json.loads(subprocess.run([tool, path, str(rank), str(world)], capture_output=True, text=True, check=True).stdout)
            lo, hi = shard_range(101, rank, world)
            assert (part["firstPair"], part["numPairs"], part["totalPairs"]) == (lo, hi - lo, 101)
            assert part["sizes"] == sizes[lo:hi] and part["fnv"] == _fnv(sb, lo, hi)
            assert part["numCells"] == sum(a * b for a, b in sizes[lo:hi])
            seen += part["numBytes"]
        assert seen == os.path.getsize(path)          # the shards partition the file's bytes
    bad = str(tmp_path / "bad.txt")
    open(bad, "wb").write(b"0\nACGT\n")
    r = subprocess.run([tool, bad, "0", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "multiple of 3" in r.stderr      # same failure mode as parseInput.cpp:38-41


def test_pack2_is_the_two_bit_image_of_the_pairs_bytes():
    """dpx_pack2 (host side of the 2-bit input, no GPU involved): base k sits in bits 2*(k%4) of byte k/4 and alphabet[code] gives the
    byte back for every position inside a pair; separators and pair numbers (outside every pair) are don't-cares; a fifth symbol
    inside a pair is refused (the caller keeps its bytes)."""
    for sb in (make_ragged_batch(40, 0, 50, 0, 70, seed=9), from_strings([("GATTACA", "GCATGCT"), ("", "ACGT"), ("T", "")]),
               from_strings([("", "")])):
        packed, alphabet = dpx.pack2(sb.sequences, sb.pairs)
        assert packed.size == (sb.sequences.size + 3) // 4 and alphabet.size == 4
        codes = np.stack([(packed >> (2 * k)) & 3 for k in range(4)], axis=1).reshape(-1)[:sb.sequences.size]
        back = alphabet[codes]
        for p in sb.pairs:
            for i, n in ((p["referenceIdx"], p["referenceSize"]), (p["queryIdx"], p["querySize"])):
                assert np.array_equal(back[i:i + n], sb.sequences[i:i + n])
    packed, alphabet = dpx.pack2(*(lambda b: (b.sequences, b.pairs))(from_strings([("GATTACA", "GCATGCT")])))
    assert bytes(alphabet) == b"GATC"  # order of first appearance, reference first
    with pytest.raises(dpx.DpxError) as e:
        five = from_strings([("ACGT", "ACGTN")])
        dpx.pack2(five.sequences, five.pairs)
    assert e.value.status == -8
    bad = from_strings([("ACGT", "ACGT")])
    bad.pairs["referenceSize"][0] = 10_000
    with pytest.raises(dpx.DpxError) as e:
        dpx.pack2(bad.sequences, bad.pairs)
    assert e.value.status == -1 or e.value.status < 0
