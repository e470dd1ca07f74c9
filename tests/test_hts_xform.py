"""hts_pack / hts_unpack / hts_rle_encode / hts_rle_decode / var_put_u64 / var_get_u64 -- the htscodecs functions
cram/cram_codecs.c calls for CRAM 4.0's E_XPACK / E_XRLE (cram_codecs.c:1399, 1520, 2106, 2278, 2103, 2276).

CPU: the oracle (oracle/hts_xform_oracle.c; parity UNPINNED, htscodecs is an absent submodule) round-trips and honours the
calling conventions the reference's call sites rely on; the header's var_* inlines equal the oracle's; libhts_bgzf.so exports
the reference-named functions.  GPU: the engine's entry points are byte-identical to the oracle in both directions, on
the same inputs, through libhtsgpu.so (context form) and libhts_bgzf.so (reference names)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import refutil

ROOT = refutil.ROOT
u8p = C.POINTER(C.c_uint8)


def _orc():
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    L.orc_hts_pack.restype = C.c_void_p
    L.orc_hts_pack.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.orc_hts_unpack.restype = C.c_void_p
    L.orc_hts_unpack.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_uint64, C.c_int, C.c_char_p]
    L.orc_hts_rle_encode.restype = C.c_void_p
    L.orc_hts_rle_encode.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.c_char_p, C.POINTER(C.c_int), C.c_char_p,
                                     C.POINTER(C.c_uint64)]
    L.orc_hts_rle_decode.restype = C.c_void_p
    L.orc_hts_rle_decode.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_uint64)]
    L.orc_var_put_u64.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
    L.orc_var_get_u64.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64)]
    return L


libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]


class Xf:
    """One calling convention over three back-ends: the oracle, libhtsgpu.so (ctx first) and libhts_bgzf.so (reference names)."""

    def __init__(self, kind, ctx=None):
        self.kind = kind
        if kind == "oracle":
            L = _orc()
            self.f = {k: getattr(L, "orc_hts_" + k) for k in ("pack", "unpack", "rle_encode", "rle_decode")}
            self.pre = ()
        elif kind == "engine":
            from htslib_amd import _native as nat
            L = nat.lib
            sig = {"pack": [C.c_void_p, C.c_char_p, C.c_int64, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64)],
                   "unpack": [C.c_void_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_uint64, C.c_int, C.c_char_p],
                   "rle_encode": [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.c_char_p, C.POINTER(C.c_int), C.c_char_p,
                                  C.POINTER(C.c_uint64)],
                   "rle_decode": [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint32, C.c_char_p,
                                  C.POINTER(C.c_uint64)]}
            self.f = {}
            for k, a in sig.items():
                fn = getattr(L, "hg_hts_" + k); fn.restype = C.c_void_p; fn.argtypes = a
                self.f[k] = fn
            self.pre = (ctx,)
        else:
            L = C.CDLL(os.path.join(ROOT, "htslib_amd", "libhts_bgzf.so"))
            o = _orc()
            self.f = {}
            for k in ("pack", "unpack", "rle_encode", "rle_decode"):
                fn = getattr(L, "hts_" + k); fn.restype = C.c_void_p; fn.argtypes = getattr(o, "orc_hts_" + k).argtypes
                self.f[k] = fn
            self.pre = ()

    def pack(self, data):
        meta = C.create_string_buffer(32); ml = C.c_int(-1); ol = C.c_uint64(0)
        p = self.f["pack"](*self.pre, bytes(data), len(data), meta, C.byref(ml), C.byref(ol))
        if not p:
            return None
        out = C.string_at(p, ol.value); libc.free(p)
        return meta.raw[:ml.value], out

    def unpack(self, data, out_len, nsym, pmap):
        out = C.create_string_buffer(max(1, out_len))
        pm = bytes(pmap) + bytes(256 - len(pmap))
        p = self.f["unpack"](*self.pre, bytes(data), len(data), out, out_len, nsym, pm)
        return out.raw[:out_len] if p else None

    def rle_encode(self, data, syms=b""):
        run = C.create_string_buffer(len(data) + 16); rl = C.c_uint64(0)
        sy = C.create_string_buffer(bytes(syms) + bytes(256 - len(syms)), 256); ns = C.c_int(len(syms)); ol = C.c_uint64(0)
        p = self.f["rle_encode"](*self.pre, bytes(data), len(data), run, C.byref(rl), sy, C.byref(ns), None, C.byref(ol))
        if not p:
            return None
        lit = C.string_at(p, ol.value); libc.free(p)
        return lit, run.raw[:rl.value], sy.raw[:ns.value]

    def rle_decode(self, lit, run, syms, cap):
        out = C.create_string_buffer(max(1, cap)); ol = C.c_uint64(cap)
        p = self.f["rle_decode"](*self.pre, bytes(lit), len(lit), bytes(run), len(run), bytes(syms), len(syms), out, C.byref(ol))
        return out.raw[:ol.value] if p else None


def cases():
    rng = np.random.default_rng(20260924)
    out = [b"", b"A", b"AAAAAAA", b"ABABABABA"]
    for nsym, n in ((1, 1000), (2, 1001), (3, 4097), (4, 77), (5, 300), (16, 70001), (17, 5000), (256, 3000)):
        out.append(rng.integers(0, nsym, n, dtype=np.uint8).tobytes() if nsym <= 17 else rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    # quality-like: few symbols, long runs (lengths crossing the 1-, 2- and 3-byte varint borders)
    q = bytearray()
    for r in (1, 2, 127, 128, 129, 500, 16383, 16384, 16385, 3, 70000, 1):
        q += bytes([int(rng.integers(33, 41))]) * r
        q += rng.integers(33, 41, int(rng.integers(0, 9)), dtype=np.uint8).tobytes()
    out.append(bytes(q))
    # a run that straddles many 64-byte steps, alone
    out.append(b"\x00" * 100000)
    out.append(b"xy" * 5000 + b"z" * 300 + b"xy" * 3)
    # flags-like series
    v = rng.choice(np.array([99, 147, 83, 163], dtype=np.uint8), 200000, p=[0.4, 0.4, 0.1, 0.1])
    out.append(v.tobytes())
    return out


def per_byte(nsym):
    return 0 if nsym <= 1 else 8 if nsym <= 2 else 4 if nsym <= 4 else 2 if nsym <= 16 else 1


def check_backend(x: Xf, ref: Xf):
    for data in cases():
        # ---- pack / unpack
        a, b = x.pack(data), ref.pack(data)
        assert a == b, f"hts_pack differs on {len(data)} bytes"
        meta, packed = b
        nsym = meta[0] if len(meta) > 1 or not data else 17
        if data:
            pb = per_byte(len(set(data)))
            got = x.unpack(packed, len(data), pb, meta[1:] if pb != 1 else b"\0")
            assert got == data, f"hts_unpack does not invert hts_pack ({len(data)} bytes, {nsym} symbols)"
            if pb >= 2 and len(data) > pb:
                assert x.unpack(packed[:-1], len(data), pb, meta[1:]) is None      # too little input
                assert ref.unpack(packed[:-1], len(data), pb, meta[1:]) is None
        # ---- rle, symbols chosen by the encoder
        a, b = x.rle_encode(data), ref.rle_encode(data)
        assert a == b, f"hts_rle_encode differs on {len(data)} bytes"
        lit, run, syms = b
        for cap in (len(data), len(data) + 100):
            assert x.rle_decode(lit, run, syms, cap) == data == ref.rle_decode(lit, run, syms, cap)
        if len(data) > 1:
            assert x.rle_decode(lit, run, syms, len(data) - 1) is None                # does not fit
            assert ref.rle_decode(lit, run, syms, len(data) - 1) is None
        if run:
            assert x.rle_decode(lit, run[:-1], syms, len(data)) is None                # run lengths run short / end in a continuation
            assert ref.rle_decode(lit, run[:-1], syms, len(data)) is None
        # ---- rle, symbols chosen by the caller (cram_xrle_encode_flush passes its own rep_score set)
        if data:
            mine = bytes(sorted(set(data)))[:3]
            a, b = x.rle_encode(data, mine), ref.rle_encode(data, mine)
            assert a == b and b[2] == mine
            assert x.rle_decode(b[0], b[1], mine, len(data)) == data


def test_oracle_conventions(built):
    o = Xf("oracle")
    meta, packed = o.pack(b"ACGTACGTAC")
    assert meta == b"\x04ACGT" and packed == bytes([0b11100100, 0b11100100, 0b0100])   # first symbol in the low bits
    assert o.unpack(packed, 10, 4, b"ACGT") == b"ACGTACGTAC"
    assert o.pack(b"aaaa") == (b"\x01a", b"")
    assert o.unpack(b"", 4, 0, b"a") == b"aaaa"
    lit, run, syms = o.rle_encode(b"aaaabcccccd")
    assert (lit, run, syms) == (b"abcd", b"\x03\x04", b"ac")                           # run length - 1, one per run of a listed symbol
    assert o.rle_decode(lit, run, syms, 64) == b"aaaabcccccd"
    lit, run, syms = o.rle_encode(b"abab")
    assert (lit, run, syms) == (b"abab", b"", b"")
    check_backend(o, o)


def test_var_u64_header_equals_oracle(built, tmp_path):
    src = tmp_path / "v.c"
    src.write_text(r'''
#include <stdio.h>
#include "hts_cram_gpu.h"
int orc_var_put_u64(uint8_t *cp, const uint8_t *endp, uint64_t v);
int orc_var_get_u64(const uint8_t *cp, const uint8_t *endp, uint64_t *v);
int main(void) {
    uint64_t vals[] = {0, 1, 127, 128, 16383, 16384, 0xffffffffull, 1ull << 35, 1ull << 56, ~0ull, 0x0123456789abcdefull};
    for (unsigned i = 0; i < sizeof vals / sizeof *vals; i++) {
        uint8_t a[16] = {0}, b[16] = {0}; uint64_t x = 1, y = 2;
        int na = var_put_u64(a, NULL, vals[i]), nb = orc_var_put_u64(b, NULL, vals[i]);
        if (na != nb || memcmp(a, b, 16)) { printf("put %u\n", i); return 1; }
        if (var_get_u64(a, a + na, &x) != na || orc_var_get_u64(b, b + nb, &y) != nb || x != vals[i] || y != vals[i]) { printf("get %u\n", i); return 1; }
        if (var_get_u64(a, NULL, &x) != na || x != vals[i]) return 1;
        if (var_put_u64(a, a + na - 1, vals[i]) != 0 || orc_var_put_u64(b, b + nb - 1, vals[i]) != 0) { printf("bound %u\n", i); return 1; }
        if (na > 1 && (var_get_u64(a, a + 1, &x) != orc_var_get_u64(b, b + 1, &y) || x != y)) { printf("short %u\n", i); return 1; }
    }
    puts("ok"); return 0; }
''')
    exe = tmp_path / "v"
    subprocess.run(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "ok"


def test_front_exports_reference_names(built):
    L = C.CDLL(os.path.join(ROOT, "htslib_amd", "libhts_bgzf.so"))
    for s in ("hts_pack", "hts_unpack", "hts_rle_encode", "hts_rle_decode"):
        assert hasattr(L, s)


@pytest.mark.gpu
def test_engine_equals_oracle(engine):
    check_backend(Xf("engine", engine._h), Xf("oracle"))


@pytest.mark.gpu
def test_reference_named_wrappers_equal_oracle(engine):
    check_backend(Xf("front"), Xf("oracle"))
