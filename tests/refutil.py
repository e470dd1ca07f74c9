"""Checker-side helpers: ctypes views of oracle/liboracle.so (our C restatement) and of the REAL
reference built in oracle/_ref (htslib bgzf.c against zlib / libdeflate).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import glob
import json
import os
import struct
import subprocess
import tempfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "bgzf")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L = self.lib
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
        L.orc_inflate_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_bgzf_uncompress_block.argtypes = [C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.orc_bgzf_decompress_stream.restype = C.c_long
        L.orc_bgzf_decompress_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_bgzf_scan.restype = C.c_long
        L.orc_bgzf_scan.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]

    def crc32(self, data: bytes, crc: int = 0) -> int:
        return self.lib.orc_crc32(crc, data, len(data))

    def inflate_raw(self, data: bytes, cap: int = 1 << 20):
        out = C.create_string_buffer(cap)
        n, used = C.c_size_t(0), C.c_size_t(0)
        rc = self.lib.orc_inflate_raw(data, len(data), out, cap, C.byref(n), C.byref(used))
        return rc, out.raw[:n.value], used.value

    def uncompress_block(self, block: bytes):
        """-> (rc, bytes) with rc as bgzf_uncompress: 0 / -1 / -2."""
        out = C.create_string_buffer(65536)
        n = C.c_size_t(65536)
        rc = self.lib.orc_bgzf_uncompress_block(out, C.byref(n), block, len(block))
        return rc, (out.raw[:n.value] if rc == 0 else b"")

    def decompress(self, stream: bytes, cap: int | None = None):
        if cap is None:
            cap = sum(b[2] for b in split_blocks(stream)) + 16
        out = C.create_string_buffer(cap)
        n = self.lib.orc_bgzf_decompress_stream(stream, len(stream), out, cap)
        return n, (out.raw[:n] if n >= 0 else b"")


def split_blocks(stream: bytes):
    """[(offset, clen, isize)] by BSIZE hopping (no validation beyond bounds)."""
    pos, out = 0, []
    while pos + 18 <= len(stream):
        bs = (stream[pos + 16] | (stream[pos + 17] << 8)) + 1
        isize = struct.unpack_from("<I", stream, pos + bs - 4)[0] if pos + bs <= len(stream) else 0
        out.append((pos, bs, isize))
        pos += bs
    return out


def golden_cases():
    man = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    for name in sorted(man):
        yield name, open(os.path.join(GOLDEN, name), "rb").read(), open(os.path.join(GOLDEN, name + ".plain"), "rb").read()


def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "ref_bgzip")) and os.path.exists(os.path.join(REF_DIR, "ref_bgzip_ld"))


def ref_bgzip(args, data: bytes, flavour: str = "zlib") -> bytes:
    """Run the real reference bgzip (oracle/_ref) on `data` through a temp file."""
    exe = os.path.join(REF_DIR, "ref_bgzip" if flavour == "zlib" else "ref_bgzip_ld")
    with tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, delete=False) as f:
        f.write(data)
        path = f.name
    try:
        r = subprocess.run([exe] + list(args) + ["-c", path], capture_output=True)
        if r.returncode != 0:
            raise RuntimeError(f"{exe} {args}: rc={r.returncode} {r.stderr[-300:]!r}")
        return r.stdout
    finally:
        os.unlink(path)


def raw_block(data: bytes, level: int = 6, strategy: int = 0, mem: int = 8) -> bytes:
    """One BGZF block around a raw-deflate payload made with given zlib parameters."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    p = co.compress(data) + co.flush()
    return wrap_payload(p, data)


def wrap_payload(payload: bytes, data: bytes, crc: int | None = None, isize: int | None = None) -> bytes:
    hdr = bytes.fromhex("1f8b08040000000000ff060042430200")
    return b"".join([hdr, struct.pack("<H", len(payload) + 25), payload,
                     struct.pack("<II", zlib.crc32(data) if crc is None else crc, len(data) if isize is None else isize)])


# ------------------------------------------------------------------ rANS 4x8 (CRAM 3.0)
RANS_GOLDEN = os.path.join(ROOT, "tests", "golden", "rans4x8")


class Rans4x8Oracle:
    def __init__(self):
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.orc_rans4x8_uncompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_rans4x8_compress.restype = C.c_size_t
        L.orc_rans4x8_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
        L.orc_rans4x8_compress_bound.restype = C.c_size_t
        L.orc_rans4x8_compress_bound.argtypes = [C.c_size_t]
        self.L = L

    def decode(self, b: bytes):
        usz = int.from_bytes(b[5:9], "little") if len(b) >= 9 else 0
        out = C.create_string_buffer(max(usz, 1))
        n = C.c_size_t(0)
        rc = self.L.orc_rans4x8_uncompress(b, len(b), out, usz, C.byref(n))
        return rc, (out.raw[:n.value] if rc == 0 else b"")

    def encode(self, d: bytes, order: int) -> bytes:
        out = C.create_string_buffer(self.L.orc_rans4x8_compress_bound(len(d)))
        n = self.L.orc_rans4x8_compress(d, len(d), out, order)
        return out.raw[:n]


def rans_golden_cases():
    man = json.load(open(os.path.join(RANS_GOLDEN, "MANIFEST.json")))
    for name in sorted(man):
        v = man[name]
        exp = bytes.fromhex(v["expected_hex"]) if v["expected_hex"] is not None else None
        if "expected_z" in v:                                      # long plaintexts travel zlib-packed beside the stream
            exp = zlib.decompress(open(os.path.join(RANS_GOLDEN, v["expected_z"]), "rb").read())
        yield name, open(os.path.join(RANS_GOLDEN, name), "rb").read(), v["usize"], exp, v["order"]


# ------------------------------------------------------------------ rANS Nx16 (CRAM 3.1) -- parity unpinned
class RansNx16Oracle:
    FLAGS = {"ORDER": 1, "X32": 4, "STRIPE": 8, "NOSZ": 16, "CAT": 32, "RLE": 64, "PACK": 128}

    def __init__(self):
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.orc_ransnx16_uncompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_ransnx16_compress.restype = C.c_size_t
        L.orc_ransnx16_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
        L.orc_ransnx16_compress_bound.restype = C.c_size_t
        L.orc_ransnx16_compress_bound.argtypes = [C.c_size_t]
        self.L = L

    def encode(self, d: bytes, flags: int) -> bytes:
        out = C.create_string_buffer(self.L.orc_ransnx16_compress_bound(len(d)))
        n = self.L.orc_ransnx16_compress(d, len(d), out, flags)
        return out.raw[:n]

    def decode(self, b: bytes, cap: int):
        out = C.create_string_buffer(max(cap, 1))
        n = C.c_size_t(0)
        rc = self.L.orc_ransnx16_uncompress(b, len(b), out, cap, C.byref(n))
        return rc, (out.raw[:n.value] if rc == 0 else b"")


class ArithOracle:
    """oracle/arith_oracle.c -- CRAM 3.1 adaptive range coder, PARITY UNPINNED."""

    def __init__(self):
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.orc_arith_uncompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_long]
        L.orc_arith_compress.restype = C.c_size_t
        L.orc_arith_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
        L.orc_arith_compress_bound.restype = C.c_size_t
        L.orc_arith_compress_bound.argtypes = [C.c_size_t]
        self.L = L

    def encode(self, d: bytes, flags: int) -> bytes:
        out = C.create_string_buffer(self.L.orc_arith_compress_bound(len(d)))
        n = self.L.orc_arith_compress(d, len(d), out, flags)
        return out.raw[:n]

    def decode(self, b: bytes, cap: int, known: int = -1):
        out = C.create_string_buffer(max(cap, 1))
        n = C.c_size_t(0)
        rc = self.L.orc_arith_uncompress(b, len(b), out, cap, C.byref(n), known)
        return rc, (out.raw[:n.value] if rc == 0 else b"")


class FqzOracle:
    """oracle/fqzcomp_oracle.c -- CRAM 3.1 fqzcomp quality codec, PARITY UNPINNED."""
    SEL, REV, DEDUP, STAB, NOQMAP = 1, 2, 4, 8, 16               # encoder options

    def __init__(self):
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.orc_fqz_encode.restype = C.c_size_t
        L.orc_fqz_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p]
        L.orc_fqz_compress_bound.restype = C.c_size_t
        L.orc_fqz_compress_bound.argtypes = [C.c_size_t, C.c_size_t]
        L.orc_fqz_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t)]
        L.orc_fqz_store_array.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.orc_fqz_read_array.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int]
        self.L = L

    def encode(self, quals: bytes, lens, rflags=None, strat: int = 0, opts: int = 0) -> bytes:
        import numpy as np
        ln = np.ascontiguousarray(lens, dtype=np.uint32)
        fl = None if rflags is None else np.ascontiguousarray(rflags, dtype=np.uint32)      # BAM flags: 16 reverse, 128 second read
        out = C.create_string_buffer(self.L.orc_fqz_compress_bound(len(quals), len(ln)))
        n = self.L.orc_fqz_encode(quals, len(quals), ln.ctypes.data, None if fl is None else fl.ctypes.data, len(ln), strat, opts, out)
        assert n > 0, "fqz oracle encoder refused the input"
        return out.raw[:n]

    def decode(self, b: bytes, cap: int, max_rec: int = 0):
        import numpy as np
        out = C.create_string_buffer(max(cap, 1))
        n, nr = C.c_size_t(0), C.c_size_t(0)
        lens = np.zeros(max(max_rec, 1), dtype=np.uint32)
        rc = self.L.orc_fqz_decode(b, len(b), out, cap, C.byref(n), lens.ctypes.data if max_rec else None, max_rec, C.byref(nr))
        return rc, (out.raw[:n.value] if rc == 0 else b""), lens[:min(nr.value, max_rec)]

    def store_array(self, a) -> bytes:
        import numpy as np
        v = np.ascontiguousarray(a, dtype=np.uint32)
        out = C.create_string_buffer(4096)
        n = self.L.orc_fqz_store_array(v.ctypes.data, len(v), out)
        return out.raw[:n]

    def read_array(self, b: bytes, size: int):
        import numpy as np
        v = np.zeros(size, dtype=np.uint32)
        return self.L.orc_fqz_read_array(b, len(b), v.ctypes.data, size), v


class Tok3Oracle:
    """oracle/tok3_oracle.c -- CRAM 3.1 read-name tokeniser, PARITY UNPINNED."""

    def __init__(self):
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.orc_tok3_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_tok3_encode.restype = C.c_size_t
        L.orc_tok3_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
        L.orc_tok3_compress_bound.restype = C.c_size_t
        L.orc_tok3_compress_bound.argtypes = [C.c_size_t]
        self.L = L

    def encode(self, d: bytes, use_arith: int = 0) -> bytes:
        out = C.create_string_buffer(self.L.orc_tok3_compress_bound(len(d)))
        n = self.L.orc_tok3_encode(d, len(d), out, use_arith)
        return out.raw[:n]

    def decode(self, b: bytes, cap: int):
        out = C.create_string_buffer(max(cap, 1))
        n = C.c_size_t(0)
        rc = self.L.orc_tok3_decode(b, len(b), out, cap, C.byref(n))
        return rc, (out.raw[:n.value] if rc == 0 else b"")
