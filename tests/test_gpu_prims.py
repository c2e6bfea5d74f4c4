"""GPU parity of the DPX primitive set (a5): the CDNA4 instruction mappings in csrc/dpx_prims.hpp, evaluated on the
device through dpx_prim_eval, against the reference's known answers (c++/testFakeDPX.cpp via
tests/golden/fakedpx_kat.json) and against the oracle on random / corner operands for all 36 entry points."""
import json
import os

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_known_answers_of_testFakeDPX(gpu):
    kat = json.load(open(os.path.join(G, "fakedpx_kat.json")))
    res, pred = gpu.prim_eval([k["op"] for k in kat], [k["a"] for k in kat], [k["b"] for k in kat], [k["c"] for k in kat])
    for k, r, p in zip(kat, res, pred):
        assert (int(r), int(p)) == (k["result"], k["pred"]), k


def test_all_36_primitives_vs_oracle(gpu):
    rng = np.random.default_rng(5)
    corner = [0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0x00FF00FF, 0xFF00FF00, 0xFFFF0001, 0x7FFF8000, 0x80007FFF, 0x00010001]
    ops, A, B, C = [], [], [], []
    for op in range(36):
        vals = corner + [int(v) for v in rng.integers(0, 1 << 32, 120, dtype=np.uint64)]
        for _ in range(300):
            a, b, c = (vals[int(i)] for i in rng.integers(0, len(vals), 3))
            ops.append(op); A.append(a); B.append(b); C.append(c)
    res, pred = gpu.prim_eval(ops, A, B, C)
    for op, a, b, c, r, p in zip(ops, A, B, C, res, pred):
        assert (int(r), int(p)) == O.dpx(op, a, b, c), (op, hex(a), hex(b), hex(c))
