"""Synthetic alignment batches and the reference's flat input layout.

The reference's golden inputs (correct-outputs/*/input-data.txt, input-datasets/) are not in the
container, so every measured workload is synthetic: reference = n bases uniform over '0'..'3'
(the reference's own alphabet, correct-outputs/LSW/web-scraper-LSW.py:7), query = the reference with
10 % substitutions, then 1 % single-base insertions and 1 % deletions, then cut / padded with random
bases to exactly m; every 97th pair is fully random (low score), every 101st pair has
query == reference prefix (maximum score).  numpy's MT19937 generator, one seed per workload.

`SynthBatch.sequences` / `.pairs` use exactly the layout the reference's parseInput() produces
(c++/parseInput.cpp:78-112): the file's bytes with every '\\n' replaced by '\\0', and per pair a
seqPair{referenceIdx, referenceSize, queryIdx, querySize} of byte offsets into that buffer.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .capi import PAIR_DTYPE

_IDX_WIDTH = 8  # line 0 of each record: zero-padded pair index (the parser ignores its content)


@dataclass
class SynthBatch:
    sequences: np.ndarray  # uint8, parseInput layout ('\0'-separated)
    pairs: np.ndarray      # PAIR_DTYPE
    m: int
    n: int

    @property
    def num_pairs(self) -> int:
        return len(self.pairs)

    @property
    def cells(self) -> int:
        return int((self.pairs["referenceSize"].astype(np.int64) * self.pairs["querySize"].astype(np.int64)).sum())

    def ref(self, p: int) -> bytes:
        r = self.pairs[p]
        return self.sequences[r["referenceIdx"]:r["referenceIdx"] + r["referenceSize"]].tobytes()

    def qry(self, p: int) -> bytes:
        r = self.pairs[p]
        return self.sequences[r["queryIdx"]:r["queryIdx"] + r["querySize"]].tobytes()


def _mutate(rng: np.random.Generator, ref: np.ndarray, m: int) -> np.ndarray:
    """ref: (n,) values 0..3 -> query of exactly m values."""
    n = len(ref)
    q = ref.copy()
    sub = rng.random(n) < 0.10
    q[sub] = (q[sub] + rng.integers(1, 4, size=int(sub.sum()))) % 4
    dele = rng.random(len(q)) < 0.01
    q = q[~dele]
    ins = np.nonzero(rng.random(len(q)) < 0.01)[0]
    if len(ins):
        q = np.insert(q, ins, rng.integers(0, 4, size=len(ins)))
    if len(q) >= m:
        return q[:m]
    return np.concatenate([q, rng.integers(0, 4, size=m - len(q))])


def make_batch(num_pairs: int, m: int, n: int, seed: int, first_index: int = 0) -> SynthBatch:
    """`num_pairs` pairs of query length m (rows) and reference length n (columns)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    refs = rng.integers(0, 4, size=(num_pairs, n), dtype=np.int64).astype(np.uint8)
    qrys = np.empty((num_pairs, m), np.uint8)
    for p in range(num_pairs):
        g = first_index + p
        if g % 97 == 96:
            qrys[p] = rng.integers(0, 4, size=m)
        elif g % 101 == 100:
            k = min(m, n)
            qrys[p, :k] = refs[p, :k]
            if m > k:
                qrys[p, k:] = rng.integers(0, 4, size=m - k)
        else:
            qrys[p] = _mutate(rng, refs[p], m)
    rec = _IDX_WIDTH + 1 + n + 1 + m + 1
    buf = np.zeros((num_pairs, rec), np.uint8)
    idx = np.arange(first_index, first_index + num_pairs, dtype=np.int64)
    for d in range(_IDX_WIDTH):
        buf[:, _IDX_WIDTH - 1 - d] = (idx // (10 ** d)) % 10 + ord("0")
    buf[:, _IDX_WIDTH + 1:_IDX_WIDTH + 1 + n] = refs + ord("0")
    buf[:, _IDX_WIDTH + 2 + n:_IDX_WIDTH + 2 + n + m] = qrys + ord("0")
    pairs = np.zeros(num_pairs, PAIR_DTYPE)
    base = np.arange(num_pairs, dtype=np.int64) * rec
    if base[-1] + rec >= 2 ** 31:
        raise ValueError("batch exceeds the reference's int32 seqPair offsets (c++/parseInput.h:22-29)")
    pairs["referenceIdx"] = base + _IDX_WIDTH + 1
    pairs["referenceSize"] = n
    pairs["queryIdx"] = base + _IDX_WIDTH + 2 + n
    pairs["querySize"] = m
    return SynthBatch(buf.reshape(-1), pairs, m, n)


def make_ragged_batch(num_pairs: int, m_lo: int, m_hi: int, n_lo: int, n_hi: int, seed: int) -> SynthBatch:
    """Short-read-like ragged batch (cfg1 of BASELINE.json: reference 100-160, query 80-130)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    chunks, pairs, off = [], np.zeros(num_pairs, PAIR_DTYPE), 0
    for p in range(num_pairs):
        n = int(rng.integers(n_lo, n_hi + 1))
        m = int(rng.integers(m_lo, m_hi + 1))
        ref = rng.integers(0, 4, size=n).astype(np.uint8)
        qry = rng.integers(0, 4, size=m).astype(np.uint8) if p % 97 == 96 else _mutate(rng, ref, m).astype(np.uint8)
        head = np.frombuffer(b"%d\0" % p, np.uint8)
        chunks += [head, ref + ord("0"), np.zeros(1, np.uint8), qry + ord("0"), np.zeros(1, np.uint8)]
        pairs[p] = (off + len(head), n, off + len(head) + n + 1, m)
        off += len(head) + n + 1 + m + 1
    return SynthBatch(np.concatenate(chunks), pairs, m_hi, n_hi)


def from_strings(pairs_text) -> SynthBatch:
    """Build the flat layout from [(reference, query), ...] python strings/bytes (small hand-made cases)."""
    chunks, pairs, off = [], np.zeros(len(pairs_text), PAIR_DTYPE), 0
    mm = nn = 0
    for p, (ref, qry) in enumerate(pairs_text):
        ref = ref.encode("latin-1") if isinstance(ref, str) else bytes(ref)
        qry = qry.encode("latin-1") if isinstance(qry, str) else bytes(qry)
        head = b"%d\0" % p
        chunks.append(head + ref + b"\0" + qry + b"\0")
        pairs[p] = (off + len(head), len(ref), off + len(head) + len(ref) + 1, len(qry))
        off += len(chunks[-1])
        mm, nn = max(mm, len(qry)), max(nn, len(ref))
    return SynthBatch(np.frombuffer(b"".join(chunks) or b"\0", np.uint8).copy(), pairs, mm, nn)


def write_pairs_file(batch: SynthBatch, path: str) -> None:
    """Write the 3-lines-per-pair text file the reference's parseInput() reads."""
    data = batch.sequences.copy()
    data[data == 0] = ord("\n")
    with open(path, "wb") as f:
        f.write(data.tobytes())


def parse_pairs_file(path: str, cap: Optional[int] = None) -> SynthBatch:
    """Python mirror of parseInput() (c++/parseInput.cpp:9-119): same buffer, same seqPair records."""
    data = np.fromfile(path, np.uint8)
    nl = np.nonzero(data == ord("\n"))[0]
    if len(nl) % 3 != 0:
        raise ValueError("Number of lines not a multiple of 3")  # parseInput.cpp:38-41 exits here
    data = data.copy()
    data[nl] = 0
    num = len(nl) // 3 if cap is None else min(cap, len(nl) // 3)
    pairs = np.zeros(num, PAIR_DTYPE)
    e0, e1, e2 = nl[0::3][:num], nl[1::3][:num], nl[2::3][:num]
    pairs["referenceIdx"] = e0 + 1
    pairs["referenceSize"] = e1 - e0 - 1
    pairs["queryIdx"] = e1 + 1
    pairs["querySize"] = e2 - e1 - 1
    m = int(pairs["querySize"].max()) if num else 0
    n = int(pairs["referenceSize"].max()) if num else 0
    return SynthBatch(data, pairs, m, n)
