"""ctypes access to the CPU oracle (oracle/liboracle.so) and, where it was built, to the real reference
classes (oracle/_ref/libref_shim.so).  TEST INFRASTRUCTURE ONLY -- never imported by the product."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_shim.so")
REF_DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
REF_DRIVER_O2 = os.path.join(ROOT, "oracle", "_ref", "ref_driver_O2")

_vp = C.c_void_p
_orc = None
_ref = None


def oracle():
    global _orc
    if _orc is None:
        lib = C.CDLL(ORACLE_SO)
        lib.orc_lsw_fill.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
        lib.orc_bsw_fill.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
        lib.orc_lnw_fill.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]
        lib.orc_anw_fill.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [_vp] * 7
        lib.orc_lsw_traceback.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
        lib.orc_lnw_traceback.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, _vp, C.c_char_p, C.c_char_p, C.c_char_p]
        lib.orc_anw_traceback.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, _vp, _vp, _vp, C.c_char_p, C.c_char_p, C.c_char_p]
        lib.orc_dpx.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.orc_dpx.restype = C.c_uint32
        lib.orc_fill_batch_timed.argtypes = [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
        lib.orc_fill_batch_timed.restype = C.c_double
        _orc = lib
    return _orc


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.ref_lsw.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_char_p, C.c_size_t]
        lib.ref_lnw.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_char_p, C.c_size_t]
        lib.ref_anw.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [_vp] * 7 + [C.c_char_p, C.c_size_t]
        lib.ref_dpx.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.ref_dpx.restype = C.c_uint32
        _ref = lib
    return _ref


def _p(a):
    return a.ctypes.data if a is not None else None


class Result:
    pass


def lsw(refs: bytes, qry: bytes, match=3, mismatch=-1, gap=-2, band=0, want_dir=True):
    n, m = len(refs), len(qry)
    r = Result()
    r.H = np.zeros((m + 1, n + 1), np.int32)
    r.dir = np.zeros((m + 1, n + 1), np.uint8) if want_dir else None
    sc, er, ec = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    if band > 0:
        oracle().orc_bsw_fill(refs, n, qry, m, match, mismatch, gap, band, _p(r.H), _p(r.dir), C.addressof(sc), C.addressof(er), C.addressof(ec))
    else:
        oracle().orc_lsw_fill(refs, n, qry, m, match, mismatch, gap, _p(r.H), _p(r.dir), C.addressof(sc), C.addressof(er), C.addressof(ec))
    r.score, r.end_row, r.end_col = sc.value, er.value, ec.value
    return r


def lnw(refs: bytes, qry: bytes, match=3, mismatch=-1, gap=-2, want_dir=True):
    n, m = len(refs), len(qry)
    r = Result()
    r.H = np.zeros((m + 1, n + 1), np.int32)
    r.dir = np.zeros((m + 1, n + 1), np.uint8) if want_dir else None
    sc = C.c_int32(0)
    oracle().orc_lnw_fill(refs, n, qry, m, match, mismatch, gap, _p(r.H), _p(r.dir), C.addressof(sc))
    r.score = sc.value
    return r


def anw(refs: bytes, qry: bytes, match=3, mismatch=-1, gap_open=-3, gap_extend=-1, want_dir=True):
    n, m = len(refs), len(qry)
    r = Result()
    r.H = np.zeros((m + 1, n + 1), np.int32)
    r.I = np.zeros((m + 1, n + 1), np.int32)
    r.D = np.zeros((m + 1, n + 1), np.int32)
    r.dirH = np.zeros((m + 1, n + 1), np.uint8) if want_dir else None
    r.dirI = np.zeros((m + 1, n + 1), np.uint8) if want_dir else None
    r.dirD = np.zeros((m + 1, n + 1), np.uint8) if want_dir else None
    sc = C.c_int32(0)
    oracle().orc_anw_fill(refs, n, qry, m, match, mismatch, gap_open, gap_extend, _p(r.H), _p(r.I), _p(r.D), _p(r.dirH), _p(r.dirI), _p(r.dirD), C.addressof(sc))
    r.score = sc.value
    return r


def _tb_bufs(m, n):
    return [C.create_string_buffer(m + n + 2) for _ in range(3)]


def lsw_traceback(refs, qry, res):
    a, b, c = _tb_bufs(len(qry), len(refs))
    k = oracle().orc_lsw_traceback(refs, len(refs), qry, len(qry), _p(res.H), _p(res.dir), res.end_row, res.end_col, a, b, c)
    assert k >= 0
    return a.raw[:k].decode("latin-1"), b.raw[:k].decode("latin-1"), c.raw[:k].decode("latin-1")  # sequences may hold NUL bytes


def lnw_traceback(refs, qry, res):
    a, b, c = _tb_bufs(len(qry), len(refs))
    k = oracle().orc_lnw_traceback(refs, len(refs), qry, len(qry), _p(res.dir), a, b, c)
    assert k >= 0
    return a.raw[:k].decode("latin-1"), b.raw[:k].decode("latin-1"), c.raw[:k].decode("latin-1")  # sequences may hold NUL bytes


def anw_traceback(refs, qry, res):
    a, b, c = _tb_bufs(len(qry), len(refs))
    k = oracle().orc_anw_traceback(refs, len(refs), qry, len(qry), _p(res.dirH), _p(res.dirI), _p(res.dirD), a, b, c)
    assert k >= 0
    return a.raw[:k].decode("latin-1"), b.raw[:k].decode("latin-1"), c.raw[:k].decode("latin-1")  # sequences may hold NUL bytes


def dpx(op, a, b, c):
    pr = C.c_uint32(0)
    r = oracle().orc_dpx(op, a & 0xFFFFFFFF, b & 0xFFFFFFFF, c & 0xFFFFFFFF, C.byref(pr))
    return r, pr.value


# ---- the real reference classes (only where oracle/_ref was built, i.e. in the build container) ----

def ref_lsw(refs: bytes, qry: bytes, match=3, mismatch=-1, gap=-2, pair_num=0):
    n, m = len(refs), len(qry)
    r = Result()
    r.H = np.zeros((m + 1, n + 1), np.int32)
    r.dir = np.zeros((m + 1, n + 1), np.uint8)
    sc = C.c_int32(0)
    text = C.create_string_buffer(4 * (m + n) + 256)
    ref().ref_lsw(refs, qry, pair_num, match, mismatch, gap, _p(r.H), _p(r.dir), C.addressof(sc), text, len(text))
    r.score, r.text = sc.value, text.value.decode("latin-1")
    return r


def ref_lnw(refs: bytes, qry: bytes, match=3, mismatch=-1, gap=-2, pair_num=0):
    n, m = len(refs), len(qry)
    r = Result()
    r.H = np.zeros((m + 1, n + 1), np.int32)
    r.dir = np.zeros((m + 1, n + 1), np.uint8)
    sc = C.c_int32(0)
    text = C.create_string_buffer(4 * (m + n) + 256)
    ref().ref_lnw(refs, qry, pair_num, match, mismatch, gap, _p(r.H), _p(r.dir), C.addressof(sc), text, len(text))
    r.score, r.text = sc.value, text.value.decode("latin-1")
    return r


def ref_anw(refs: bytes, qry: bytes, match=3, mismatch=-1, gap_open=-3, gap_extend=-1, pair_num=0):
    n, m = len(refs), len(qry)
    r = Result()
    r.H, r.I, r.D = (np.zeros((m + 1, n + 1), np.int32) for _ in range(3))
    r.dirH, r.dirI, r.dirD = (np.zeros((m + 1, n + 1), np.uint8) for _ in range(3))
    sc = C.c_int32(0)
    text = C.create_string_buffer(4 * (m + n) + 256)
    ref().ref_anw(refs, qry, pair_num, match, mismatch, gap_open, gap_extend, _p(r.H), _p(r.I), _p(r.D), _p(r.dirH), _p(r.dirI), _p(r.dirD), C.addressof(sc), text, len(text))
    r.score, r.text = sc.value, text.value.decode("latin-1")
    return r


def ref_dpx(op, a, b, c):
    pr = C.c_uint32(0)
    r = ref().ref_dpx(op, a & 0xFFFFFFFF, b & 0xFFFFFFFF, c & 0xFFFFFFFF, C.byref(pr))
    return r, pr.value
