"""GPU drop-in tests of the C++ host mirror (dpx_gpu_genomics_project_amd/hostcpp):

* oracle/_ref/main_dropin_{LSW,LNW,ANW} -- the reference's OWN c++/main.cpp, compiled unchanged against the mirror's
  headers and libdpxalign.so (hostcpp/Makefile `dropin`; built in the container that has /root/reference, the binary
  travels to the GPU box) -- must print, after reordering by pair number (its 20 pthreads print in any order, which
  is what scripts/reorderOutput.py exists for), exactly the blocks the reference's CPU classes print.
* dpx_class_main (same shape, run-time algorithm flag, tail fixed) and dpx_main (batched, pipelined GPU driver).
"""
import gzip
import os
import re
import subprocess

import pytest

import oracle_py as O
from dpx_gpu_genomics_project_amd.synth import make_ragged_batch, write_pairs_file

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
HOST = os.path.join(ROOT, "dpx_gpu_genomics_project_amd", "hostcpp")
W = {"LSW": ["-match", "3", "-mismatch", "-1", "-open", "-2"], "LNW": ["-match", "3", "-mismatch", "-1", "-open", "-2"],
     "ANW": ["-match", "3", "-mismatch", "-1", "-open", "-3", "-extend", "-1"]}


def golden(algo):
    return gzip.open(os.path.join(G, f"short400_{algo}.out.gz"), "rb").read().decode("latin-1")


def blocks_sorted(stdout):
    """Split a driver's stdout into (header, {pair: block}, footer); a block starts at '<n> | <score>' and has 4 lines."""
    lines = stdout.split("\n")
    out, i, header, footer = {}, 0, [], []
    while i < len(lines):
        m = re.match(r"^(\d+) \| (-?\d+)$", lines[i])
        if m:
            out[int(m.group(1))] = "\n".join(lines[i:i + 4]) + "\n"
            i += 4
        else:
            (footer if out else header).append(lines[i])
            i += 1
    return header, out, footer


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, encoding="latin-1", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    return r.stdout


@pytest.mark.parametrize("algo", ["LSW", "LNW", "ANW"])
def test_reference_main_cpp_drops_onto_the_engine(algo):
    exe = os.path.join(ROOT, "oracle", "_ref", f"main_dropin_{algo}")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/main_dropin_* not built (needs /root/reference at build time)")
    out = run([exe, "-pairs", os.path.join(G, "short400.txt")] + W[algo])
    header, blocks, footer = blocks_sorted(out)
    assert header[0].startswith("Parsing input file: ") and header[1] == "Pair # | Score"
    assert any(l.startswith("Elapsed time (usec): ") for l in footer) and "Cleaning up" in footer
    assert sorted(blocks) == list(range(400))
    assert "".join(blocks[p] for p in range(400)) == golden(algo)


def test_reference_main_cpp_with_several_concurrent_leaders(monkeypatch):
    """hostcpp/DpxPair.cpp deals the batches it forms from main.cpp's 20 threads to one leader per visible device;
    DPX_CLASS_LEADERS=3 rehearses three concurrent leaders (all on this box's one GPU): same blocks."""
    exe = os.path.join(ROOT, "oracle", "_ref", "main_dropin_LNW")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/main_dropin_* not built (needs /root/reference at build time)")
    monkeypatch.setenv("DPX_CLASS_LEADERS", "3")
    _, blocks, _ = blocks_sorted(run([exe, "-pairs", os.path.join(G, "short400.txt")] + W["LNW"]))
    assert "".join(blocks[p] for p in range(400)) == golden("LNW")
    _, blocks, _ = blocks_sorted(run([os.path.join(HOST, "dpx_class_main"), "-pairs", os.path.join(G, "short400.txt")] + W["ANW"] + ["-algo", "ANW"]))
    assert "".join(blocks[p] for p in range(400)) == golden("ANW")


@pytest.mark.parametrize("algo", ["LSW", "LNW", "ANW"])
def test_class_driver_and_batched_driver_match_reference_stdout(algo):
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    pairs = os.path.join(G, "short400.txt")
    _, blocks, _ = blocks_sorted(run([os.path.join(HOST, "dpx_class_main"), "-pairs", pairs] + W[algo] + ["-algo", algo]))
    assert "".join(blocks[p] for p in range(400)) == golden(algo)
    # batched driver prints in input order; 3 uneven batches exercise the printer pipeline
    out = run([os.path.join(HOST, "dpx_main"), "-pairs", pairs] + W[algo] + ["-algo", algo, "-batch", "150"])
    start = out.index("Pair # | Score\n") + len("Pair # | Score\n")
    end = out.index("Elapsed time (usec): ")
    assert out[start:end] == golden(algo)
    assert re.search(r"^GCUPS: \d+\.\d+$", out, re.M) and "Num Pairs: 400" in out
    # 11 batches: create / fill / traceback of batch k+1 are issued while batch k is in flight and batch k-1 is printing
    out = run([os.path.join(HOST, "dpx_main"), "-pairs", pairs] + W[algo] + ["-algo", algo, "-batch", "37"])
    assert out[out.index("Pair # | Score\n") + 15:out.index("Elapsed time (usec): ")] == golden(algo)


def test_batches_from_the_pool_budget_print_the_same_text(tmp_path):
    """Round 3: without -batch the batched driver sizes its batches from a matrix-pool budget (-pool-gb; two pools are reserved on a
    helper thread while the file is parsed, dpx_pool_reserve).  100 pairs of 700 x 900 under a 64-MiB budget = 3 batches on two
    recycled pools: same stdout as one explicit batch, scores as the oracle."""
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    from dpx_gpu_genomics_project_amd.synth import make_batch
    sb = make_batch(100, 700, 900, seed=31, first_index=90)
    path = str(tmp_path / "p100.txt")
    write_pairs_file(sb, path)
    body = lambda out: out[out.index("Pair # | Score\n") + 15:out.index("Elapsed time (usec): ")]
    for algo in ("LSW", "ANW"):
        one = body(run([os.path.join(HOST, "dpx_main"), "-pairs", path] + W[algo] + ["-algo", algo, "-batch", "100"]))
        auto = body(run([os.path.join(HOST, "dpx_main"), "-pairs", path] + W[algo] + ["-algo", algo, "-pool-gb", "0.0625"]))
        dflt = body(run([os.path.join(HOST, "dpx_main"), "-pairs", path] + W[algo] + ["-algo", algo]))
        assert one == auto == dflt
        # the producer thread (default for batches of >= 8192 pairs) issues batch k+1 while batch k is waited for: same text, batch by batch
        for prod in ("0", "1"):
            assert body(run([os.path.join(HOST, "dpx_main"), "-pairs", path] + W[algo] + ["-algo", algo, "-batch", "7", "-producer", prod])) == one
        if algo == "LSW":
            for p in (0, 41, 42, 99):
                assert f"\n{p} | {O.lsw(sb.ref(p), sb.qry(p), 3, -1, -2, want_dir=False).score}\n" in "\n" + one


def test_pipeline_depth_producer_threads_and_tuned_pools_print_the_same_text(tmp_path):
    """Round 4: -inflight K batches on the device at a time, -producer P threads creating them (finished in input order), -tune 1 (the
    first batch on every pool shops for its placement: DPX_TUNE_PLACEMENT on pools of >= 1 GiB that dpx_pool_reserve built ahead of time)."""
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    from dpx_gpu_genomics_project_amd.synth import make_batch
    sb = make_batch(1300, 1024, 1024, seed=33, first_index=95)     # 1-GiB budget: ~430 pairs per batch, 4 batches
    path = str(tmp_path / "p1300.txt")
    write_pairs_file(sb, path)
    body = lambda out: out[out.index("Pair # | Score\n") + 15:out.index("Elapsed time (usec): ")]
    base = [os.path.join(HOST, "dpx_main"), "-pairs", path] + W["LSW"] + ["-algo", "LSW", "-pool-gb", "1"]
    want = body(run(base))
    for extra in (["-inflight", "3"], ["-inflight", "1"], ["-producer", "2"], ["-producer", "3", "-inflight", "2"], ["-tune", "1"], ["-tune", "1", "-producer", "2"]):
        assert body(run(base + extra)) == want, extra
    for p in (0, 431, 1299):
        assert f"\n{p} | {O.lsw(sb.ref(p), sb.qry(p), 3, -1, -2, want_dir=False).score}\n" in "\n" + want


def test_tail_pairs_are_not_dropped(tmp_path):
    """The reference main drops pairs past the last full 400 (c++/main.cpp:169); the mirror's drivers must not."""
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    sb = make_ragged_batch(37, 30, 60, 40, 70, seed=12)
    path = str(tmp_path / "p37.txt")
    write_pairs_file(sb, path)
    want = ""
    for p in range(37):
        o = O.lsw(sb.ref(p), sb.qry(p), 3, -1, -2)
        a, b, c = ("", "", "") if o.score == 0 else O.lsw_traceback(sb.ref(p), sb.qry(p), o)
        want += f"{p} | {o.score}\n{a}\n{b}\n{c}\n"
    _, blocks, _ = blocks_sorted(run([os.path.join(HOST, "dpx_class_main"), "-pairs", path] + W["LSW"]))
    assert "".join(blocks[p] for p in range(37)) == want
    out = run([os.path.join(HOST, "dpx_main"), "-pairs", path] + W["LSW"] + ["-algo", "BSW", "-band", "1000"])
    assert out[out.index("Pair # | Score\n") + 15:out.index("Elapsed time")] == want


def test_reference_testFakeDPX_passes_on_device_primitives():
    """The reference's own c++/testFakeDPX.cpp (74 asserts), compiled unchanged against hostcpp/FakeDPX.hpp, whose every
    call runs the CDNA4 mapping of the primitive on the GPU (dpx_prim_eval)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "testFakeDPX_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/testFakeDPX_dropin not built (needs /root/reference at build time)")
    out = run([exe])
    assert "PASSED ALL ASSERTIONS FOR INSTRUCTION CHECKING!!" in out


def test_sharded_ranks_concatenate_to_the_reference_stdout():
    """tools/run_multi_gpu.sh: 3 dpx_main ranks (all on device 0 here), each aligning its own shard of the file; the
    concatenation of their blocks in rank order is the reference's output for the whole file."""
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    env = dict(os.environ, DPX_SHARE_GPU="1")
    r = subprocess.run([os.path.join(ROOT, "tools", "run_multi_gpu.sh"), "3", "-pairs", os.path.join(G, "short400.txt")] + W["ANW"] +
                       ["-algo", "ANW", "-batch", "70"], capture_output=True, text=True, encoding="latin-1", env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == golden("ANW")
    assert "[rank 2] Rank 2 of 3: pairs [268, 400)" in r.stderr


@pytest.mark.parametrize("algo", ["LSW", "LNW", "MULTINW", "ANW"])
def test_host_backtrackers_over_exported_matrices_match_reference_stdout(algo):
    """hostcpp/backtrack.cpp (the reference's output plumbing, c++/backtrack.cpp:21-356): matrices exported with
    dpx_batch_matrix -> dpxDirectionsFromScores -> backtrackSW / NW / MultiNW / ANW must print the reference's blocks."""
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    gold = "LNW" if algo == "MULTINW" else algo
    out = run([os.path.join(HOST, "backtrack_test"), "-pairs", os.path.join(G, "short400.txt"), "-algo", algo] + W[gold])
    assert out[out.index("0 | "):] == golden(gold)


def test_host_print_matrix_helpers(tmp_path):
    """printMatrix / printBacktrackMatrix (c++/backtrack.cpp:3-19: ' %4d ' per cell, one line per row) on an exported matrix."""
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    from dpx_gpu_genomics_project_amd.synth import from_strings
    sb = from_strings([("0123012", "01301")])
    path = str(tmp_path / "one.txt")
    write_pairs_file(sb, path)
    out = run([os.path.join(HOST, "backtrack_test"), "-pairs", path, "-algo", "LNW", "-dump", "0"] + W["LNW"])
    o = O.lnw(sb.ref(0), sb.qry(0), 3, -1, -2)
    fmt = lambda M: "".join("".join(" %4d " % v for v in row) + "\n" for row in M)
    # direction numbering is the reference's enum (c++/backtrack.h:14-20), which is also the oracle's
    assert out[out.index(" "):] == fmt(o.H) + fmt(o.dir)
