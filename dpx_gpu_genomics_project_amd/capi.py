"""ctypes binding of include/dpx_align.h (libdpxalign.so).

Every function here goes straight through the C ABI; nothing is computed in Python.  The library is
built in-tree by ``__graft_entry__.build()`` (``make -C dpx_gpu_genomics_project_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

ALGO_LNW, ALGO_LSW, ALGO_ANW, ALGO_BSW = 0, 1, 2, 3
ALGO_NAMES = {ALGO_LNW: "LNW", ALGO_LSW: "LSW", ALGO_ANW: "ANW", ALGO_BSW: "BSW"}
KEEP_MATRICES, SCORE_ONLY, TIME_FILLS, TUNE_PLACEMENT = 0x0, 0x1, 0x2, 0x4
MAT_H, MAT_I, MAT_D = 0, 1, 2

# every symbol include/dpx_align.h declares (tests check the .so exports all of them)
ABI_VERSION_NEEDED = 3  # include/dpx_align.h DPX_ABI_VERSION: round-3 entry points (dpx_pool_reserve, dpx_batch_last_output_usec) + the pool record in dpx_batch_describe

ABI_SYMBOLS = (
    "dpx_init", "dpx_device_count", "dpx_device_info", "dpx_shutdown", "dpx_pool_reserve", "dpx_text_reserve", "dpx_strerror", "dpx_last_error",
    "dpx_abi_version", "dpx_batch_create", "dpx_batch_create_on", "dpx_pack2", "dpx_batch_create_packed2", "dpx_batch_fill", "dpx_batch_fill_timed", "dpx_batch_last_fill_usec", "dpx_batch_last_output_usec", "dpx_batch_sync",
    "dpx_batch_device_results", "dpx_batch_results", "dpx_batch_matrix", "dpx_batch_traceback",
    "dpx_batch_output_begin", "dpx_batch_output_end", "dpx_batch_output_take", "dpx_text_free",
    "dpx_batch_info", "dpx_batch_describe", "dpx_batch_destroy", "dpx_align_batch", "dpx_prim_eval",
)


class DpxError(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__(f"{what}: status {status}")
        self.status = status


class SeqPair(C.Structure):
    """== struct seqPair of the reference (c++/parseInput.h:22-29)."""
    _fields_ = [("referenceIdx", C.c_int32), ("referenceSize", C.c_int32), ("queryIdx", C.c_int32),
                ("querySize", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("algo", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32), ("gapOpen", C.c_int32),
                ("gapExtend", C.c_int32), ("band", C.c_int32)]


PAIR_DTYPE = np.dtype([("referenceIdx", "<i4"), ("referenceSize", "<i4"), ("queryIdx", "<i4"), ("querySize", "<i4")])

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    """In-tree library; DPX_LIB may point at an experimental build (tools/ A-B runs)."""
    return os.environ.get("DPX_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdpxalign.so")


def load() -> C.CDLL:
    """Load libdpxalign.so (no GPU needed just to load it).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise DpxError(-2, f"{path} is missing -- run __graft_entry__.build() (no CPU fallback exists)")
    lib = C.CDLL(path)
    vp, i32p, u32p, i16p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int16)
    lib.dpx_init.argtypes = [C.c_int]
    lib.dpx_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.dpx_device_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    lib.dpx_pool_reserve.argtypes = [C.c_size_t, C.c_int]
    lib.dpx_text_reserve.argtypes = [C.c_size_t, C.c_int]
    lib.dpx_strerror.argtypes = [C.c_int]
    lib.dpx_strerror.restype = C.c_char_p
    lib.dpx_last_error.restype = C.c_char_p
    lib.dpx_batch_create.argtypes = [C.POINTER(Params), vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_uint,
                                     C.POINTER(vp)]
    lib.dpx_batch_create_on.argtypes = [C.c_int, C.POINTER(Params), vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_uint,
                                        C.POINTER(vp)]
    lib.dpx_pack2.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, vp]
    lib.dpx_batch_create_packed2.argtypes = [C.c_int, C.POINTER(Params), vp, C.c_size_t, vp, vp, C.c_size_t, C.c_size_t, C.c_uint,
                                             C.POINTER(vp)]
    lib.dpx_batch_fill.argtypes = [vp, vp]
    lib.dpx_batch_fill_timed.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.dpx_batch_last_fill_usec.argtypes = [vp, C.POINTER(C.c_double)]
    lib.dpx_batch_last_output_usec.argtypes = [vp, C.POINTER(C.c_double)]
    lib.dpx_batch_sync.argtypes = [vp]
    lib.dpx_batch_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.dpx_batch_results.argtypes = [vp, vp, vp, vp]
    lib.dpx_batch_matrix.argtypes = [vp, C.c_size_t, C.c_int, vp]
    lib.dpx_batch_traceback.argtypes = [vp, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p, i32p]
    lib.dpx_batch_output_begin.argtypes = [vp, C.c_uint64]
    lib.dpx_batch_output_end.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint64))]
    lib.dpx_batch_output_take.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.dpx_text_free.argtypes = [vp]
    lib.dpx_batch_info.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64)]
    lib.dpx_batch_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.dpx_batch_destroy.argtypes = [vp]
    lib.dpx_align_batch.argtypes = [C.POINTER(Params), vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, vp, vp, vp]
    lib.dpx_prim_eval.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, vp]
    for name in ABI_SYMBOLS:  # every declared entry point must be exported
        getattr(lib, name)
    if lib.dpx_abi_version() < ABI_VERSION_NEEDED:
        raise DpxError(-8, f"{path} has ABI version {lib.dpx_abi_version()}, this binding needs >= {ABI_VERSION_NEEDED} -- rebuild it")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = lib.dpx_strerror(rc).decode()
        detail = lib.dpx_last_error().decode()
        raise DpxError(rc, f"{what}: {msg}" + (f" [{detail}]" if detail else ""))


def init(device: int = 0) -> None:
    _check(load().dpx_init(device), "dpx_init")


def device_count() -> int:
    n = C.c_int(0)
    rc = load().dpx_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def device_info() -> Tuple[str, int, int]:
    name = C.create_string_buffer(256)
    cus, mem = C.c_int(0), C.c_size_t(0)
    _check(load().dpx_device_info(name, 256, C.byref(cus), C.byref(mem)), "dpx_device_info")
    return name.value.decode(), cus.value, mem.value


def pack2(sequences: np.ndarray, pairs: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """dpx_pack2 (host only, no GPU): (packed uint8[(n+3)//4], alphabet uint8[4]) of a flat byte buffer whose pairs use at most four
    byte values; DpxError(-8) otherwise."""
    lib = load()
    seq = np.ascontiguousarray(sequences, dtype=np.uint8)
    prs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    packed = np.zeros((seq.size + 3) // 4, np.uint8)
    alphabet = np.zeros(4, np.uint8)
    _check(lib.dpx_pack2(seq.ctypes.data, seq.size, prs.ctypes.data, len(prs), alphabet.ctypes.data, packed.ctypes.data), "dpx_pack2")
    return packed, alphabet


class Batch:
    """A device-resident batch of alignment pairs (dpx_batch).  `packed2=(packed, alphabet, num_bases)`: the sequences arrive as 2-bit
    codes (dpx_batch_create_packed2) and `sequences` is ignored."""

    def __init__(self, algo: int, sequences: np.ndarray, pairs: np.ndarray, match: int = 3, mismatch: int = -1,
                 gap_open: int = -2, gap_extend: int = -1, band: int = 0, flags: int = KEEP_MATRICES,
                 first_pair: int = 0, num_pairs: Optional[int] = None, device: int = -1, packed2=None):
        lib = load()
        self._lib = lib
        self._h = C.c_void_p(None)
        seq = np.ascontiguousarray(sequences if sequences is not None else np.zeros(0, np.uint8), dtype=np.uint8)
        prs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        if num_pairs is None:
            num_pairs = len(prs) - first_pair
        self.params = Params(algo, match, mismatch, gap_open, gap_extend, band)
        self.pairs = prs[first_pair:first_pair + num_pairs].copy()
        self.num_pairs = num_pairs
        if packed2 is not None:
            pk = np.ascontiguousarray(packed2[0], dtype=np.uint8)
            al = np.ascontiguousarray(packed2[1], dtype=np.uint8)
            rc = lib.dpx_batch_create_packed2(device, C.byref(self.params), pk.ctypes.data, int(packed2[2]), al.ctypes.data, prs.ctypes.data,
                                              first_pair, num_pairs, flags, C.byref(self._h))
            _check(rc, "dpx_batch_create_packed2")
            return
        rc = lib.dpx_batch_create_on(device, C.byref(self.params), seq.ctypes.data, seq.size, prs.ctypes.data, first_pair,
                                     num_pairs, flags, C.byref(self._h))
        _check(rc, "dpx_batch_create")

    def fill(self, stream: int = 0) -> None:
        _check(self._lib.dpx_batch_fill(self._h, C.c_void_p(stream) if stream else None), "dpx_batch_fill")

    def fill_timed(self, repeats: int = 1) -> float:
        """Mean device microseconds of one fill over `repeats` back-to-back launches (hipEvents)."""
        us = C.c_double(0.0)
        _check(self._lib.dpx_batch_fill_timed(self._h, repeats, C.byref(us)), "dpx_batch_fill_timed")
        return us.value

    def sync(self) -> None:
        _check(self._lib.dpx_batch_sync(self._h), "dpx_batch_sync")

    def results(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        s = np.empty(self.num_pairs, np.int32)
        r = np.empty(self.num_pairs, np.int32)
        c = np.empty(self.num_pairs, np.int32)
        _check(self._lib.dpx_batch_results(self._h, s.ctypes.data, r.ctypes.data, c.ctypes.data), "dpx_batch_results")
        return s, r, c

    def device_results(self) -> Tuple[int, int, int]:
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(self._lib.dpx_batch_device_results(self._h, C.byref(a), C.byref(b), C.byref(c)),
               "dpx_batch_device_results")
        return a.value, b.value, c.value

    def matrix(self, pair: int, which: int = MAT_H) -> np.ndarray:
        m, n = int(self.pairs["querySize"][pair]), int(self.pairs["referenceSize"][pair])
        out = np.empty((m + 1, n + 1), np.int16)
        _check(self._lib.dpx_batch_matrix(self._h, pair, which, out.ctypes.data), "dpx_batch_matrix")
        return out

    def traceback(self, pair: int) -> Tuple[str, str, str]:
        m, n = int(self.pairs["querySize"][pair]), int(self.pairs["referenceSize"][pair])
        cap = m + n + 2
        a, b, c = (C.create_string_buffer(cap) for _ in range(3))
        ln = C.c_int32(0)
        _check(self._lib.dpx_batch_traceback(self._h, pair, a, b, c, C.byref(ln)), "dpx_batch_traceback")
        k = ln.value
        return a.raw[:k].decode("latin-1"), b.raw[:k].decode("latin-1"), c.raw[:k].decode("latin-1")

    def output_begin(self, first_pair_number: int = 0) -> None:
        """Start building the batch's result text on the device (asynchronous)."""
        _check(self._lib.dpx_batch_output_begin(self._h, first_pair_number), "dpx_batch_output_begin")

    def output_end(self) -> Tuple[bytes, np.ndarray]:
        """(text, offsets): the reference's stdout blocks of every pair, and numPairs + 1 byte offsets into it."""
        text, nbytes, offs = C.c_char_p(), C.c_size_t(0), C.POINTER(C.c_uint64)()
        _check(self._lib.dpx_batch_output_end(self._h, C.byref(text), C.byref(nbytes), C.byref(offs)), "dpx_batch_output_end")
        raw = C.string_at(text, nbytes.value)
        return raw, np.ctypeslib.as_array(offs, shape=(self.num_pairs + 1,)).copy()

    def info(self) -> dict:
        npairs, cells, mb, ab = C.c_size_t(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(self._lib.dpx_batch_info(self._h, C.byref(npairs), C.byref(cells), C.byref(mb), C.byref(ab)),
               "dpx_batch_info")
        return {"num_pairs": npairs.value, "cells": cells.value, "matrix_bytes": mb.value,
                "algorithmic_bytes": ab.value}

    def describe(self) -> dict:
        """How the engine fills this batch (dpx_batch_describe): kernel, arithmetic type, launch-list sizes."""
        buf = C.create_string_buffer(2048)
        _check(self._lib.dpx_batch_describe(self._h, buf, 2048), "dpx_batch_describe")
        out = dict(kv.split("=", 1) for kv in buf.value.decode().split())
        return {k: (int(v) if v.lstrip("-").isdigit() else v) for k, v in out.items()}

    def close(self) -> None:
        if self._h and self._h.value:
            self._lib.dpx_batch_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prim_eval(ops: Sequence[int], a: Sequence[int], b: Sequence[int], c: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """Evaluate DPX primitives on the device (dpx_prim_eval).  Returns (result, pred)."""
    op = np.ascontiguousarray(ops, np.int32)
    A = np.ascontiguousarray(np.asarray(a, np.uint64) & 0xFFFFFFFF, np.uint32)
    B = np.ascontiguousarray(np.asarray(b, np.uint64) & 0xFFFFFFFF, np.uint32)
    Cc = np.ascontiguousarray(np.asarray(c, np.uint64) & 0xFFFFFFFF, np.uint32)
    res = np.empty(len(op), np.uint32)
    pred = np.empty(len(op), np.uint32)
    _check(load().dpx_prim_eval(op.ctypes.data, A.ctypes.data, B.ctypes.data, Cc.ctypes.data, len(op),
                                res.ctypes.data, pred.ctypes.data), "dpx_prim_eval")
    return res, pred
