"""CRAM block methods 2 (bzip2) and 3 (lzma): cram_uncompress_block of the drop-in front-end hands them to the system's libbz2 / liblzma,
as the reference does (cram/cram_io.c:1626-1664) -- there is no GPU form of either.  Runs without a GPU: these blocks never reach the
engine.  The expected plaintext comes from Python's bz2 / lzma modules, which bind the same libraries' formats."""
import bz2, ctypes as C, lzma, os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CramBlock(C.Structure):               # = struct cram_block (include/hts_cram_gpu.h, cram/cram_structs.h:312-332)
    _fields_ = [("method", C.c_int), ("orig_method", C.c_int), ("content_type", C.c_int), ("content_id", C.c_int32), ("comp_size", C.c_int32),
                ("uncomp_size", C.c_int32), ("crc32", C.c_uint32), ("idx", C.c_int32), ("data", C.c_void_p), ("alloc", C.c_size_t), ("byte", C.c_size_t),
                ("bit", C.c_int), ("m", C.c_void_p), ("crc32_checked", C.c_int), ("crc_part", C.c_uint32)]


@pytest.fixture(scope="module")
def front(built):
    L = C.CDLL(os.path.join(ROOT, "htslib_amd", "libhts_bgzf.so"))
    L.cram_uncompress_block.argtypes = [C.POINTER(CramBlock)]
    return L


def make_block(libc, method, payload, usize):
    b = CramBlock()
    b.method = method; b.orig_method = 0; b.content_type = 4; b.content_id = 7; b.comp_size = len(payload); b.uncomp_size = usize
    p = libc.malloc(max(len(payload), 1)); C.memmove(p, payload, len(payload))      # the block owns malloc'd data (cram_io.c:1615-1617)
    b.data = p; b.alloc = len(payload); b.crc32_checked = 1                           # the CRC was checked by the reader (its device batch is covered elsewhere)
    return b


def test_bzip2_and_lzma_blocks_go_to_the_system_libraries(front):
    libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(2)
    plain = bytes(rng.integers(0, 4, 200_000, dtype=np.uint8)) + b"ACGT" * 5000
    for method, comp in ((2, bz2.compress(plain, 9)), (3, lzma.compress(plain, format=lzma.FORMAT_XZ))):
        b = make_block(libc, method, comp, len(plain))
        assert front.cram_uncompress_block(C.byref(b)) == 0
        assert b.method == 0 and b.alloc == len(plain) and C.string_at(b.data, len(plain)) == plain     # RAW now, data replaced
        libc.free(b.data)
        # wrong declared size / damaged payload: -1 and the block untouched (the reference's size check, cram_io.c:1639-1642, 1655-1658)
        for bad_payload, usz in ((comp, len(plain) - 1), (comp, len(plain) + 1), (comp[:len(comp) // 2], len(plain)), (comp[:20] + bytes(40) + comp[60:], len(plain))):
            b = make_block(libc, method, bad_payload, usz)
            assert front.cram_uncompress_block(C.byref(b)) == -1
            assert b.method == method and C.string_at(b.data, len(bad_payload)) == bad_payload
            libc.free(b.data)
