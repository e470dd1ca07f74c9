"""ctypes view of include/hts_bgzf_gpu.h (htslib_amd/libhts_bgzf.so) for the front-end tests."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class BGZF(C.Structure):
    _fields_ = [("errcode", C.c_uint, 16), ("reserved", C.c_uint, 1), ("is_write", C.c_uint, 1),
                ("no_eof_block", C.c_uint, 1), ("is_be", C.c_uint, 1), ("compress_level", C.c_int, 9),
                ("last_block_eof", C.c_uint, 1), ("is_compressed", C.c_uint, 1), ("is_gzip", C.c_uint, 1),
                ("cache_size", C.c_int), ("block_length", C.c_int), ("block_clength", C.c_int), ("block_offset", C.c_int),
                ("block_address", C.c_int64), ("uncompressed_address", C.c_int64),
                ("uncompressed_block", C.c_void_p), ("compressed_block", C.c_void_p), ("cache", C.c_void_p),
                ("fp", C.c_void_p), ("mt", C.c_void_p), ("idx", C.c_void_p), ("idx_build_otf", C.c_int),
                ("gz_stream", C.c_void_p), ("seeked", C.c_int64)]


class KString(C.Structure):
    _fields_ = [("l", C.c_size_t), ("m", C.c_size_t), ("s", C.c_void_p)]


def load():
    L = C.CDLL(os.path.join(ROOT, "htslib_amd", "libhts_bgzf.so"))
    P = C.POINTER(BGZF)
    L.bgzf_open.restype = P; L.bgzf_open.argtypes = [C.c_char_p, C.c_char_p]
    L.bgzf_dopen.restype = P; L.bgzf_dopen.argtypes = [C.c_int, C.c_char_p]
    L.bgzf_close.argtypes = [P]
    L.bgzf_read.restype = C.c_ssize_t; L.bgzf_read.argtypes = [P, C.c_void_p, C.c_size_t]
    L.bgzf_write.restype = C.c_ssize_t; L.bgzf_write.argtypes = [P, C.c_char_p, C.c_size_t]
    L.bgzf_flush.argtypes = [P]; L.bgzf_flush_try.argtypes = [P, C.c_ssize_t]
    L.bgzf_seek.restype = C.c_int64; L.bgzf_seek.argtypes = [P, C.c_int64, C.c_int]
    L.bgzf_getc.argtypes = [P]; L.bgzf_peek.argtypes = [P]
    L.bgzf_getline.argtypes = [P, C.c_int, C.POINTER(KString)]
    L.bgzf_check_EOF.argtypes = [P]; L.bgzf_read_block.argtypes = [P]
    L.bgzf_mt.argtypes = [P, C.c_int, C.c_int]
    L.bgzf_compress.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.c_int]
    L.bgzf_index_build_init.argtypes = [P]
    L.bgzf_index_dump.argtypes = [P, C.c_char_p, C.c_char_p]; L.bgzf_index_load.argtypes = [P, C.c_char_p, C.c_char_p]
    L.bgzf_useek.restype = C.c_int; L.bgzf_useek.argtypes = [P, C.c_long, C.c_int]
    L.bgzf_utell.restype = C.c_long; L.bgzf_utell.argtypes = [P]
    L.bgzf_is_bgzf.argtypes = [C.c_char_p]; L.bgzf_compression.argtypes = [P]
    L.hts_crc32.restype = C.c_uint32; L.hts_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
    L.bgzf_hfile.restype = C.c_void_p; L.bgzf_hfile.argtypes = [P]
    return L


def tell(fp):
    return (fp.contents.block_address << 16) | (fp.contents.block_offset & 0xFFFF)


def read_all(L, fp, chunk=100_000):
    out = []
    buf = C.create_string_buffer(chunk)
    while True:
        n = L.bgzf_read(fp, buf, chunk)
        if n < 0:
            raise IOError(f"bgzf_read failed errcode={fp.contents.errcode}")
        if n == 0:
            break
        out.append(buf.raw[:n])
    return b"".join(out)
