"""SURVEY 8 row a15 under htscodecs' own NAMES: the entry points htslib calls in htscodecs (cram/cram_io.c:1668-1891), exported by libhts_bgzf.so
(htslib_amd/csrc/htscodecs_front.cpp) with htscodecs' signatures and ownership -- what an htslib built --with-external-htscodecs links in place of
libhtscodecs.so.  Each call is one stream on the engine.  Checked against oracle/'s restatements: the bytes a compress function returns are the oracle's for the
same flags (rANS 4x8: the PINNED restatement), the oracle decodes them, and the uncompress functions return the input.  The same functions run inside
oracle/_ref/libhts_gpu.so under the reference's own callers (tests/test_libhts_gpu.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import refutil

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hc(engine):
    L = C.CDLL(os.path.join(refutil.ROOT, "htslib_amd", "libhts_bgzf.so"))
    vp, up = C.c_void_p, C.POINTER(C.c_uint)
    for name, args in (("rans_compress", [C.c_char_p, C.c_uint, up, C.c_int]), ("rans_uncompress", [C.c_char_p, C.c_uint, up]),
                       ("rans_compress_4x16", [C.c_char_p, C.c_uint, up, C.c_int]), ("rans_uncompress_4x16", [C.c_char_p, C.c_uint, up]),
                       ("arith_compress_to", [C.c_char_p, C.c_uint, vp, up, C.c_int]), ("arith_uncompress_to", [C.c_char_p, C.c_uint, vp, up]),
                       ("tok3_encode_names", [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), vp]), ("tok3_decode_names", [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]),
                       ("fqz_compress", [C.c_int, vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, vp]),
                       ("fqz_decompress", [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_int])):
        f = getattr(L, name); f.restype = vp; f.argtypes = args
    L.htscodecs_version.restype = C.c_char_p
    L.free = C.CDLL(None).free; L.free.argtypes = [vp]
    return L


def take(L, p, n):
    assert p, "NULL result"
    b = C.string_at(p, n); L.free(p)
    return b


def test_every_codec_under_its_htscodecs_name(hc):
    rng = np.random.default_rng(4)
    assert b"gfx950" in hc.htscodecs_version()
    qual = bytes(np.clip(38 + np.cumsum(rng.integers(-1, 2, 120_000)) % 10, 2, 41).astype(np.uint8))
    small = bytes(rng.integers(0, 5, 3000, dtype=np.uint8))
    n = C.c_uint(0)
    # rANS 4x8 (pinned restatement)
    r4 = refutil.Rans4x8Oracle()
    for d in (qual, small):
        for order in (0, 1):
            c = take(hc, hc.rans_compress(d, len(d), C.byref(n), order), n.value)
            assert c == r4.encode(d, order)
            assert take(hc, hc.rans_uncompress(c, len(c), C.byref(n)), n.value) == d
    # rANS Nx16: RANS_ORDER_SIMD_AUTO (0x8000) -> 32-way from 64 KiB
    nx = refutil.RansNx16Oracle()
    for d in (qual, small):
        for fl in (0, 1, 64, 9, 128, 193):
            c = take(hc, hc.rans_compress_4x16(d, len(d), C.byref(n), fl | 0x8000), n.value)
            assert c == nx.encode(d, fl | (4 if len(d) >= 65536 else 0)), (len(d), fl)
            assert take(hc, hc.rans_uncompress_4x16(c, len(c), C.byref(n)), n.value) == d
    # adaptive range coder
    ar = refutil.ArithOracle()
    for d in (qual, small):
        for fl in (0, 1, 64, 65, 193):
            c = take(hc, hc.arith_compress_to(d, len(d), None, C.byref(n), fl), n.value)
            assert c == ar.encode(d, fl), (len(d), fl)
            assert take(hc, hc.arith_uncompress_to(c, len(c), None, C.byref(n)), n.value) == d
    # name tokeniser
    tk = refutil.Tok3Oracle()
    names = b"".join(b"SIM:1:FC01:%d:%d:%d:%d\0" % (1 + i % 8, 1101 + i % 1577, 1000 + (i * 7919) % 20000, 2000 + (i * 104729) % 30000) for i in range(3000))
    for ua in (0, 1):
        m = C.c_int(0)
        c = take(hc, hc.tok3_encode_names(names, len(names), 3, ua, C.byref(m), None), m.value)
        rc, back = tk.decode(c, len(names))
        assert rc == 0 and back == names and len(c) < len(names) // 3
        u = C.c_uint32(0)
        assert take(hc, hc.tok3_decode_names(c, len(c), C.byref(u)), u.value) == names
    # fqzcomp: fqz_slice { int num_records; uint32_t *len; uint32_t *flags; }
    class Slice(C.Structure):
        _fields_ = [("num_records", C.c_int), ("len", C.POINTER(C.c_uint32)), ("flags", C.POINTER(C.c_uint32))]
    nrec, rl = 800, 150
    q = qual[:nrec * rl]
    lens = (C.c_uint32 * nrec)(*([rl] * nrec)); flags = (C.c_uint32 * nrec)(*[16 if i % 3 == 0 else 0 for i in range(nrec)])
    sl = Slice(nrec, lens, flags)
    fq = refutil.FqzOracle()
    for strat in (0, 1, 2, 3):
        cs = C.c_size_t(0)
        c = take(hc, hc.fqz_compress(3, C.byref(sl), q, len(q), C.byref(cs), strat, None), cs.value)
        want = fq.encode(q, np.full(nrec, rl, np.uint32), np.array([16 if i % 3 == 0 else 0 for i in range(nrec)], np.uint32), strat,
                         fq.DEDUP | fq.REV | (fq.SEL if strat == 1 else 0))
        assert c == want, strat
        us = C.c_size_t(0)
        assert take(hc, hc.fqz_decompress(c, len(c), C.byref(us), None, 0), us.value) == q
