"""ctypes binding of the C ABI in include/htsgpu.h (libhtsgpu.so, built in-tree).

There is no Python or CPU implementation behind these calls: if the shared
library has not been built (``make`` / ``__graft_entry__.build()``) importing
this module raises, and if no gfx950 device is usable ``Engine()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HTSGPU_LIB") or os.path.join(_HERE, "libhtsgpu.so")   # (HTSGPU_LIB: an instrumented build of the same library, for probes)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the gfx950 engine first (run `make` at the repo root or "
        "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")

lib = C.CDLL(LIB_PATH)


class BgzfDesc(C.Structure):
    """struct hg_bgzf_desc (include/htsgpu.h)."""
    _fields_ = [("coff", C.c_uint64), ("uoff", C.c_uint64), ("clen", C.c_uint32), ("ulen", C.c_uint32)]


DESC_DTYPE = [("coff", "<u8"), ("uoff", "<u8"), ("clen", "<u4"), ("ulen", "<u4")]

_vp = C.c_void_p
lib.hg_version.restype = C.c_char_p
lib.hg_strerror.restype = C.c_char_p
lib.hg_strerror.argtypes = [C.c_int]
lib.hg_init.argtypes = [C.c_int, C.POINTER(_vp)]
lib.hg_destroy.argtypes = [_vp]
lib.hg_destroy.restype = None
lib.hg_device_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.hg_bgzf_scan.restype = C.c_long
lib.hg_bgzf_scan.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_uint64)]
lib.hg_bgzf_inflate_dev.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]
lib.hg_bgzf_inflate_host.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t),
                                     _vp, C.c_size_t, C.POINTER(C.c_long), C.POINTER(C.c_int)]
lib.hg_crc32_dev.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]
lib.hg_bgzf_deflate_dev.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp, _vp, _vp]
lib.hg_bgzf_pack_dev.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, C.c_int, _vp]
lib.hg_bgzf_deflate_host.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int, _vp, C.c_size_t,
                                     C.POINTER(C.c_size_t)]

lib.hg_rans4x8_decode_dev.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp]
lib.hg_rans4x8_decode_host.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp]

lib.hg_gzip_inflate_dev.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]
lib.hg_cram_uncompress_blocks_host.argtypes = [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp]

lib.hg_ransnx16_decode_host.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]
lib.hg_ransnx16_decode_dev.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, _vp]

lib.hg_ransnx16_compress_bound.restype = C.c_size_t
lib.hg_ransnx16_compress_bound.argtypes = [C.c_size_t]
lib.hg_ransnx16_encode_host.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]

lib.hg_rans4x8_compress_bound.restype = C.c_size_t
lib.hg_rans4x8_compress_bound.argtypes = [C.c_size_t]
lib.hg_rans4x8_encode_host.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]

lib.hg_gzip_compress_bound.restype = C.c_size_t
lib.hg_gzip_compress_bound.argtypes = [C.c_size_t]
lib.hg_cram_compress_bound.restype = C.c_size_t
lib.hg_cram_compress_bound.argtypes = [C.c_size_t]
lib.hg_gzip_deflate_host.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp, _vp]
lib.hg_cram_compress_blocks_host.argtypes = [_vp, C.c_size_t, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]

lib.hg_arith_compress_bound.restype = C.c_size_t
lib.hg_arith_compress_bound.argtypes = [C.c_size_t]
lib.hg_arith_decode_host.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]
lib.hg_arith_encode_host.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]
lib.hg_fqz_decode_host.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]
lib.hg_cram_decode_bam_host.argtypes = [_vp, C.c_size_t, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_uint64, _vp, C.c_size_t, _vp, _vp, _vp, _vp]
lib.hg_cram_file_to_bam_host.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_int, _vp, C.c_size_t, _vp, _vp]
lib.hg_cram_file_to_bam_host2.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_int, _vp, C.c_size_t, _vp, _vp, C.c_int, C.c_char_p]
lib.hg_cram_decode_bam_host2.argtypes = [_vp, C.c_size_t, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_uint64, _vp, C.c_size_t, _vp, _vp, _vp, _vp, C.c_char_p]
lib.hg_cram_encode_slices_host.argtypes = [_vp, _vp, C.c_size_t, C.c_size_t, C.c_uint32, _vp, C.c_int, _vp, C.c_int, C.c_int64, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]
lib.hg_bam_to_cram_host.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_int, C.c_uint32, C.c_int, _vp, C.c_size_t, _vp, _vp]
lib.hg_bam_to_cram_host2.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_int, C.c_uint32, C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp]
lib.hg_cram_index_build_host.restype = C.c_long
lib.hg_cram_index_build_host.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_size_t]
lib.hg_cram_batch_stage.argtypes = [_vp, C.c_size_t, _vp, C.c_int, C.c_int, C.c_uint64, _vp]
lib.hg_cram_batch_decode_bam_dev.argtypes = [_vp, _vp, _vp, C.c_int, C.c_char_p, _vp, _vp, _vp, _vp, _vp]
lib.hg_cram_batch_read_bam.argtypes = [_vp, _vp, _vp, C.c_size_t]
lib.hg_cram_batch_free.argtypes = [_vp, _vp]
lib.hg_cram_batch_free.restype = None
lib.hg_cram_crai_slice.restype = C.c_long
lib.hg_cram_crai_slice.argtypes = [_vp, C.c_uint32, C.c_int, _vp, _vp, _vp, C.c_int64, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]
lib.hg_cram_records_bound.argtypes = [C.c_size_t, _vp, C.c_int, _vp, _vp, _vp, _vp]
lib.hg_cram_decode_records_host.argtypes = [_vp, C.c_size_t, _vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp, _vp]
lib.hg_fqz_compress_bound.restype = C.c_size_t
lib.hg_fqz_compress_bound.argtypes = [C.c_size_t, C.c_size_t]
lib.hg_fqz_encode_host.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]


class FqzSlice(C.Structure):
    """hg_fqz_slice: the reference's fqz_slice (record lengths + BAM flags of one QS block, cram_io.c:1808-1820)."""
    _fields_ = [("num_records", C.c_uint32), ("len", _vp), ("flags", _vp)]

lib.hg_tok3_decode_host.argtypes = [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp]

lib.hg_tok3_compress_bound.restype = C.c_size_t
lib.hg_tok3_compress_bound.argtypes = [C.c_size_t]
lib.hg_tok3_encode_host.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp]

lib.hg_cram_metrics_new.restype = _vp
lib.hg_cram_metrics_new.argtypes = []
lib.hg_cram_metrics_free.argtypes = [_vp]
lib.hg_cram_compress_blocks_metrics_host.argtypes = [_vp, C.c_size_t, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
lib.hg_cram_compress_blocks_metrics_fqz_host.argtypes = [_vp, C.c_size_t, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]


class CramSliceBlocks(C.Structure):
    """hg_cram_slice_blocks"""
    _fields_ = [("comp_hdr", _vp), ("comp_hdr_len", C.c_uint32), ("slice_hdr", _vp), ("slice_hdr_len", C.c_uint32), ("core", _vp), ("core_len", C.c_uint32),
                ("nblocks", C.c_uint32), ("content_id", _vp), ("data", _vp), ("len", _vp), ("nrefs", C.c_uint32), ("refs", _vp), ("decode_md", C.c_int32)]


class CramRefSpan(C.Structure):
    """hg_cram_ref_span"""
    _fields_ = [("ref_id", C.c_int32), ("start", C.c_int64), ("bases", _vp), ("len", C.c_uint32), ("sq_len", C.c_int64)]


def cram_slice_array(slices, keep, decode_md=-1, with_refs=True):
    """slice dicts {"comp_hdr", "slice_hdr", "core", "blocks": [(content id, bytes)], "refs": [(ref id, start, bases, @SQ length)]} -> ctypes array of
    hg_cram_slice_blocks; `keep` receives the buffers that must stay alive.  Slices of one container may share their compression header bytes object."""
    import numpy as np
    arr = (CramSliceBlocks * len(slices))()
    same = {}
    for i, s in enumerate(slices):
        ch = same.setdefault(s["comp_hdr"], C.create_string_buffer(s["comp_hdr"], len(s["comp_hdr"])))
        sh = C.create_string_buffer(s["slice_hdr"], len(s["slice_hdr"])); co = C.create_string_buffer(s["core"], max(len(s["core"]), 1))
        bl = [C.create_string_buffer(d, max(len(d), 1)) for _, d in s["blocks"]]
        ids = np.array([cid for cid, _ in s["blocks"]], dtype=np.int32); lens = np.array([len(d) for _, d in s["blocks"]], dtype=np.uint32)
        ptrs = (_vp * max(len(bl), 1))(*[C.addressof(x) for x in bl])
        refs = s.get("refs", []) if with_refs else []
        rb = [C.create_string_buffer(b, max(len(b), 1)) for _, _, b, _ in refs]
        ra = (CramRefSpan * max(len(rb), 1))(*[CramRefSpan(t, a, C.addressof(buf), len(b), ln) for (t, a, b, ln), buf in zip(refs, rb)])
        keep.append((ch, sh, co, bl, ids, lens, ptrs, rb, ra))
        arr[i] = CramSliceBlocks(C.addressof(ch), len(s["comp_hdr"]), C.addressof(sh), len(s["slice_hdr"]), C.addressof(co), len(s["core"]), len(bl), ids.ctypes.data,
                                 C.addressof(ptrs), lens.ctypes.data, len(rb), C.addressof(ra), decode_md)
    return arr


class CramMetrics(C.Structure):
    """struct hg_cram_metrics (= the reference's struct cram_metrics)."""
    _fields_ = [("trial", C.c_int), ("next_trial", C.c_int), ("consistency", C.c_int), ("sz", C.c_int * 32),
                ("input_avg_sz", C.c_int), ("input_avg_delta", C.c_int), ("method", C.c_int), ("revised_method", C.c_int),
                ("strat", C.c_int), ("cnt", C.c_int * 32), ("extra", C.c_double * 32), ("unpackable", C.c_int)]


lib.hg_bam_header_host.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
lib.hg_bam_frame_dev.restype = C.c_long
lib.hg_bam_frame_dev.argtypes = [_vp, _vp, C.c_uint64, C.c_uint64, C.c_int32, _vp, C.c_uint64, C.POINTER(C.c_uint64), _vp]
lib.hg_bam_bases_dev.argtypes = [_vp, _vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64), _vp]

lib.hg_bam_core_dev.argtypes = [_vp, _vp, _vp, C.c_uint64, _vp, _vp]
lib.hg_bam_quals_dev.argtypes = [_vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp]


class BamCoreCols(C.Structure):
    _fields_ = [(k, _vp) for k in ("tid", "pos", "bin", "mapq", "l_qname", "flag", "n_cigar", "l_qseq", "mtid", "mpos", "isize")]


lib.hg_bai_build_dev.restype = C.c_long
lib.hg_bai_build_dev.argtypes = [_vp, _vp, C.c_uint64, C.c_uint64, C.c_int32, _vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint64, _vp, C.c_size_t, _vp]

lib.hg_idx_build_dev.restype = C.c_long
lib.hg_idx_build_dev.argtypes = [_vp, _vp, C.c_uint64, C.c_uint64, C.c_int32, _vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                 C.c_int, _vp, C.c_size_t, _vp]
lib.hg_csi_levels.argtypes = [C.c_uint64, C.c_int]

lib.hg_cram_uncompress_blocks_crc_host.argtypes = [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]

EXPORTS = ["hg_version", "hg_strerror", "hg_init", "hg_destroy", "hg_device_info", "hg_bgzf_scan",
           "hg_bgzf_inflate_dev", "hg_bgzf_inflate_host", "hg_crc32_dev", "hg_bgzf_deflate_dev", "hg_bgzf_pack_dev",
           "hg_bgzf_deflate_host", "hg_rans4x8_decode_dev", "hg_rans4x8_decode_host",
           "hg_gzip_inflate_dev", "hg_cram_uncompress_blocks_host", "hg_ransnx16_decode_host", "hg_ransnx16_decode_dev",
           "hg_ransnx16_compress_bound", "hg_ransnx16_encode_host", "hg_rans4x8_compress_bound",
           "hg_rans4x8_encode_host", "hg_gzip_compress_bound", "hg_gzip_deflate_host", "hg_cram_compress_bound",
           "hg_cram_compress_blocks_host", "hg_arith_decode_host", "hg_arith_compress_bound", "hg_arith_encode_host", "hg_cram_records_bound", "hg_cram_crai_slice", "hg_cram_file_to_bam_host", "hg_cram_decode_bam_host", "hg_cram_decode_records_host", "hg_cram_batch_stage", "hg_cram_batch_decode_bam_dev", "hg_cram_batch_read_bam", "hg_cram_batch_free", "hg_cram_file_to_bam_host2", "hg_cram_containers_to_bam_host", "hg_cram_writer_new", "hg_cram_writer_free", "hg_cram_writer_containers_host", "hg_cram_decode_bam_host2", "hg_cram_encode_slices_host", "hg_cram_encode_slices_host2", "hg_bam_to_cram_host", "hg_bam_to_cram_host2", "hg_cram_index_build_host", "hg_fqz_decode_host", "hg_fqz_compress_bound", "hg_fqz_encode_host", "hg_tok3_decode_host", "hg_tok3_compress_bound", "hg_tok3_encode_host", "hg_cram_metrics_new", "hg_cram_metrics_free",
           "hg_cram_compress_blocks_metrics_host", "hg_cram_compress_blocks_metrics_fqz_host", "hg_bam_header_host", "hg_bam_frame_dev", "hg_bam_bases_dev", "hg_bam_core_dev", "hg_bam_quals_dev", "hg_bai_build_dev", "hg_idx_build_dev", "hg_csi_levels", "hg_cram_uncompress_blocks_crc_host",
           "hg_pipe_create", "hg_pipe_destroy", "hg_pipe_input", "hg_pipe_reserve", "hg_pipe_inflate", "hg_pipe_deflate", "hg_pipe_wait",
           "hg_gzip_stream_inflate_host", "hg_crc32_host", "hg_crc32_batch_host",
           "hg_hts_pack", "hg_hts_unpack", "hg_hts_rle_encode", "hg_hts_rle_decode",
           "hg_cram_itf8_decode_dev", "hg_cram_itf8_encode_dev", "hg_cram_itf8_decode_host", "hg_cram_itf8_encode_host",
           "hg_cram_byte_array_stop_dev", "hg_cram_byte_array_stop_host"]


class HgError(RuntimeError):
    def __init__(self, code: int, what: str = ""):
        self.code = code
        super().__init__(f"{what}: {lib.hg_strerror(code).decode()} ({code})")


def check(rc: int, what: str = "htsgpu"):
    if rc != 0:
        raise HgError(rc, what)


def bgzf_scan(buf) -> "tuple":
    """Host framing scan -> (numpy structured array of descriptors, total_ulen)."""
    import numpy as np
    mv = memoryview(buf)
    arr = (C.c_char * len(mv)).from_buffer_copy(mv) if mv.readonly else (C.c_char * len(mv)).from_buffer(mv)
    total = C.c_uint64(0)
    n = lib.hg_bgzf_scan(C.addressof(arr), len(mv), None, 0, C.byref(total))
    if n < 0:
        raise HgError(int(n), "hg_bgzf_scan")
    desc = np.zeros(int(n), dtype=DESC_DTYPE)
    if n:
        lib.hg_bgzf_scan(C.addressof(arr), len(mv), desc.ctypes.data, int(n), C.byref(total))
    return desc, int(total.value)


class Engine:
    """One engine context bound to one MI355X (hg_ctx)."""

    def __init__(self, device: int = 0):
        h = _vp()
        check(lib.hg_init(device, C.byref(h)), "hg_init")
        self._h = h
        self.device = device
        cus, waves = C.c_int(), C.c_int()
        lib.hg_device_info(h, C.byref(cus), C.byref(waves))
        self.cus, self.waves = cus.value, waves.value

    def close(self):
        if self._h:
            lib.hg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-buffer convenience (synchronous) --------------------------------
    def bgzf_inflate_host(self, comp: bytes):
        """Returns (plain bytes, per-block status ndarray). Raises HgError on failure."""
        import numpy as np
        desc, total = bgzf_scan(comp)
        out = C.create_string_buffer(max(total, 1))
        status = np.full(len(desc), 99, dtype=np.int32)
        out_len = C.c_size_t(0)
        bad_i, bad_c = C.c_long(-1), C.c_int(0)
        src = (C.c_char * len(comp)).from_buffer_copy(comp)
        rc = lib.hg_bgzf_inflate_host(self._h, C.addressof(src), len(comp), C.addressof(out), total,
                                      C.byref(out_len), status.ctypes.data, len(status),
                                      C.byref(bad_i), C.byref(bad_c))
        self.last_status = status
        self.last_bad = (bad_i.value, bad_c.value)
        check(rc, f"hg_bgzf_inflate_host (block {bad_i.value} code {bad_c.value})")
        return out.raw[:out_len.value], status

    def bgzf_deflate_host(self, plain: bytes, level: int = 6, cuts=None, add_eof: bool = True) -> bytes:
        """BGZF-compress `plain` on the GPU; blocks end at `cuts` (default every 0xff00 bytes)."""
        import numpy as np
        n = len(plain)
        nb = (len(cuts) - 1) if cuts is not None else (n + 0xFF00 - 1) // 0xFF00
        cap = nb * 65536 + 64
        out = C.create_string_buffer(cap)
        out_len = C.c_size_t(0)
        src = (C.c_char * max(n, 1)).from_buffer_copy(plain if n else b"\0")
        carr = None
        if cuts is not None:
            carr = np.ascontiguousarray(np.asarray(cuts, dtype=np.uint64))
        rc = lib.hg_bgzf_deflate_host(self._h, C.addressof(src), n, carr.ctypes.data if carr is not None else None,
                                      nb if carr is not None else 0, level, 1 if add_eof else 0, C.addressof(out), cap,
                                      C.byref(out_len))
        check(rc, "hg_bgzf_deflate_host")
        return out.raw[:out_len.value]

    def rans4x8_decode_host(self, streams):
        """Decode a list of CRAM 3.0 rANS 4x8 streams (bytes) -> (list of bytes, status ndarray)."""
        import numpy as np
        n = len(streams)
        if n == 0:
            return [], np.zeros(0, dtype=np.int32)
        ins = [(C.c_char * max(len(s), 1)).from_buffer_copy(s if len(s) else b"\0") for s in streams]
        in_ptr = (_vp * n)(*[C.addressof(b) for b in ins])
        in_len = np.array([len(s) for s in streams], dtype=np.uint32)
        usz = [int.from_bytes(s[5:9], "little") if len(s) >= 9 else 0 for s in streams]
        outs = [C.create_string_buffer(max(u, 1)) for u in usz]
        out_ptr = (_vp * n)(*[C.addressof(b) for b in outs])
        out_cap = np.array([max(u, 1) for u in usz], dtype=np.uint32)
        out_len = np.zeros(n, dtype=np.uint32)
        status = np.full(n, 99, dtype=np.int32)
        rc = lib.hg_rans4x8_decode_host(self._h, in_ptr, in_len.ctypes.data, n, out_ptr, out_cap.ctypes.data,
                                        out_len.ctypes.data, status.ctypes.data)
        self.last_status = status
        if rc not in (0, -6):
            check(rc, "hg_rans4x8_decode_host")
        return [outs[i].raw[:int(out_len[i])] for i in range(n)], status

    def cram_uncompress_blocks(self, blocks):
        """blocks: list of (method, comp_bytes, uncomp_size) as found in CRAM block headers.
        Returns (list of plaintext bytes or None, status ndarray) -- the batch form of
        cram_uncompress_block (cram/cram_io.c:1576-1754)."""
        import numpy as np
        n = len(blocks)
        if n == 0:
            return [], np.zeros(0, dtype=np.int32)
        ins = [(C.c_char * max(len(b[1]), 1)).from_buffer_copy(b[1] if len(b[1]) else b"\0") for b in blocks]
        outs = [C.create_string_buffer(max(b[2], 1)) for b in blocks]
        in_ptr = (_vp * n)(*[C.addressof(x) for x in ins])
        out_ptr = (_vp * n)(*[C.addressof(x) for x in outs])
        method = np.array([b[0] for b in blocks], dtype=np.int32)
        in_len = np.array([len(b[1]) for b in blocks], dtype=np.uint32)
        out_len = np.array([b[2] for b in blocks], dtype=np.uint32)
        status = np.full(n, 99, dtype=np.int32)
        rc = lib.hg_cram_uncompress_blocks_host(self._h, n, method.ctypes.data, in_ptr, in_len.ctypes.data, out_ptr,
                                                out_len.ctypes.data, status.ctypes.data)
        if rc not in (0, -6):
            check(rc, "hg_cram_uncompress_blocks_host")
        return [outs[i].raw[:blocks[i][2]] if status[i] == 0 else None for i in range(n)], status

    def cram_uncompress_blocks_crc(self, blocks):
        """blocks: (method, comp_bytes, uncomp_size, crc_part, crc32) -- with cram_uncompress_block's CRC check."""
        import numpy as np
        n = len(blocks)
        ins = [(C.c_char * max(len(b[1]), 1)).from_buffer_copy(b[1] if len(b[1]) else b"\0") for b in blocks]
        outs = [C.create_string_buffer(max(b[2], 1)) for b in blocks]
        in_ptr = (_vp * n)(*[C.addressof(x) for x in ins]); out_ptr = (_vp * n)(*[C.addressof(x) for x in outs])
        method = np.array([b[0] for b in blocks], dtype=np.int32)
        in_len = np.array([len(b[1]) for b in blocks], dtype=np.uint32); out_len = np.array([b[2] for b in blocks], dtype=np.uint32)
        part = np.array([b[3] for b in blocks], dtype=np.uint32); crc = np.array([b[4] for b in blocks], dtype=np.uint32)
        status = np.full(n, 99, dtype=np.int32)
        rc = lib.hg_cram_uncompress_blocks_crc_host(self._h, n, method.ctypes.data, in_ptr, in_len.ctypes.data, part.ctypes.data, crc.ctypes.data,
                                                    out_ptr, out_len.ctypes.data, status.ctypes.data)
        if rc not in (0, -6):
            check(rc, "hg_cram_uncompress_blocks_crc_host")
        return [outs[i].raw[:blocks[i][2]] if status[i] == 0 else None for i in range(n)], status

    def _ptr_batch(self, datas, bound):
        import numpy as np
        n = len(datas)
        ins = [(C.c_char * max(len(d), 1)).from_buffer_copy(d if len(d) else b"\0") for d in datas]
        outs = [C.create_string_buffer(bound(len(d))) for d in datas]
        return (ins, outs, (_vp * n)(*[C.addressof(x) for x in ins]), (_vp * n)(*[C.addressof(x) for x in outs]),
                np.array([len(d) for d in datas], dtype=np.uint32), np.zeros(n, dtype=np.uint32))

    def gzip_deflate_host(self, datas, level=6):
        """Each bytes object -> one gzip member (CRAM GZIP block payload)."""
        if not datas:
            return []
        ins, outs, ip, op, il, ol = self._ptr_batch(datas, lib.hg_gzip_compress_bound)
        check(lib.hg_gzip_deflate_host(self._h, ip, il.ctypes.data, len(datas), level, op, ol.ctypes.data), "hg_gzip_deflate_host")
        return [outs[i].raw[:int(ol[i])] for i in range(len(datas))]

    def cram_compress_blocks(self, datas, masks, level=5):
        """Trial-compress each block with every method in its mask; -> (list of payloads, method ids)."""
        import numpy as np
        if not datas:
            return [], np.zeros(0, dtype=np.int32)
        ins, outs, ip, op, il, ol = self._ptr_batch(datas, lib.hg_cram_compress_bound)
        mk = np.array(masks, dtype=np.uint32)
        used = np.full(len(datas), -9, dtype=np.int32)
        check(lib.hg_cram_compress_blocks_host(self._h, len(datas), mk.ctypes.data, level, ip, il.ctypes.data, op,
                                               ol.ctypes.data, used.ctypes.data), "hg_cram_compress_blocks_host")
        return [outs[i].raw[:int(ol[i])] for i in range(len(datas))], used

    def cram_compress_blocks_metrics(self, datas, metrics, method_sets, level=5, version_major=3, fqz=None):
        """cram_compress_block with the auto-tuner: metrics[i] = pointer from hg_cram_metrics_new (or None);
        fqz[i] = None or (record lengths, BAM flags or None) of a quality block -- the cram_slice argument of cram_compress_block2."""
        import numpy as np
        if not datas:
            return [], np.zeros(0, dtype=np.int32)
        ins, outs, ip, op, il, ol = self._ptr_batch(datas, lib.hg_cram_compress_bound)
        mk = np.array(method_sets, dtype=np.uint32)
        mp = (_vp * len(datas))(*[m if m else None for m in metrics])
        used = np.full(len(datas), -9, dtype=np.int32)
        keep, slp = [], None
        if fqz is not None:
            ptrs = []
            for f in fqz:
                if f is None:
                    ptrs.append(None); continue
                ln = np.ascontiguousarray(f[0], dtype=np.uint32)
                fl = None if f[1] is None else np.ascontiguousarray(f[1], dtype=np.uint32)
                sl = FqzSlice(len(ln), ln.ctypes.data, None if fl is None else fl.ctypes.data)
                keep.append((ln, fl, sl)); ptrs.append(C.addressof(sl))
            slp = (_vp * len(datas))(*ptrs)
        check(lib.hg_cram_compress_blocks_metrics_fqz_host(self._h, len(datas), mp, mk.ctypes.data, level, version_major, ip,
                                                           il.ctypes.data, slp, op, ol.ctypes.data, used.ctypes.data),
              "hg_cram_compress_blocks_metrics_fqz_host")
        return [outs[i].raw[:int(ol[i])] for i in range(len(datas))], used

    def rans4x8_encode_host(self, datas, orders):
        """rANS 4x8-encode each bytes object with the matching order (0/1) -> list of streams."""
        import numpy as np
        n = len(datas)
        if n == 0:
            return []
        ins = [(C.c_char * max(len(d), 1)).from_buffer_copy(d if len(d) else b"\0") for d in datas]
        outs = [C.create_string_buffer(lib.hg_rans4x8_compress_bound(len(d))) for d in datas]
        in_ptr = (_vp * n)(*[C.addressof(x) for x in ins])
        out_ptr = (_vp * n)(*[C.addressof(x) for x in outs])
        in_len = np.array([len(d) for d in datas], dtype=np.uint32)
        od = np.array(orders, dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.uint32)
        check(lib.hg_rans4x8_encode_host(self._h, in_ptr, in_len.ctypes.data, od.ctypes.data, n, out_ptr,
                                         out_len.ctypes.data), "hg_rans4x8_encode_host")
        return [outs[i].raw[:int(out_len[i])] for i in range(n)]

    def arith_encode_host(self, datas, flags):
        """Adaptive-range-code each bytes object in `datas` with the matching flag byte -> list of streams."""
        import numpy as np
        if not datas:
            return []
        ins, outs, ip, op, il, ol = self._ptr_batch(datas, lib.hg_arith_compress_bound)
        fl = np.array(flags, dtype=np.uint8)
        check(lib.hg_arith_encode_host(self._h, ip, il.ctypes.data, fl.ctypes.data, len(datas), op, ol.ctypes.data), "hg_arith_encode_host")
        return [outs[i].raw[:int(ol[i])] for i in range(len(datas))]

    def fqz_encode_host(self, datas, lens, flags, strats):
        """fqzcomp-encode each quality buffer; lens[i] / flags[i] = its records' lengths / BAM flags (flags[i] may be None),
        strats[i] = 0..3 -> list of CRAM method-7 payloads (b"" = refused)."""
        import numpy as np
        n = len(datas)
        if n == 0:
            return []
        ins = [(C.c_char * max(len(d), 1)).from_buffer_copy(d if len(d) else b"\0") for d in datas]
        ln = [np.ascontiguousarray(x, dtype=np.uint32) for x in lens]
        fl = [None if x is None else np.ascontiguousarray(x, dtype=np.uint32) for x in flags]
        outs = [C.create_string_buffer(lib.hg_fqz_compress_bound(len(d), len(l))) for d, l in zip(datas, ln)]
        sl = [FqzSlice(len(l), l.ctypes.data, None if f is None else f.ctypes.data) for l, f in zip(ln, fl)]
        slp = (_vp * n)(*[C.addressof(x) for x in sl])
        ip = (_vp * n)(*[C.addressof(x) for x in ins]); op = (_vp * n)(*[C.addressof(x) for x in outs])
        il = np.array([len(d) for d in datas], dtype=np.uint32); ol = np.zeros(n, dtype=np.uint32)
        st = np.array(strats, dtype=np.int32)
        check(lib.hg_fqz_encode_host(self._h, ip, il.ctypes.data, slp, st.ctypes.data, n, op, ol.ctypes.data), "hg_fqz_encode_host")
        return [outs[i].raw[:int(ol[i])] for i in range(n)]

    def cram_decode_bam(self, slice_array, nslices, major, nref, rg_names, total_bases, cap, name_prefix=None):
        """hg_cram_decode_bam_host on a cram_slice_array: -> (uncompressed BAM bytes as ndarray view, record offsets per slice, status)"""
        import numpy as np
        out = np.empty(cap, np.uint8); rec_off = np.zeros(nslices + 1, np.uint64); st = np.full(nslices, 9, np.int32); total = C.c_uint64()
        rg = [r.encode() if isinstance(r, str) else r for r in rg_names]
        rgp = (C.c_char_p * max(len(rg), 1))(*rg) if rg else None
        rc = lib.hg_cram_decode_bam_host2(self._h, nslices, C.cast(slice_array, _vp), major, nref, C.cast(rgp, _vp) if rg else None, len(rg), total_bases,
                                          out.ctypes.data, cap, rec_off.ctypes.data, None, C.byref(total), st.ctypes.data, name_prefix)
        if rc not in (0, -6):
            check(rc, "hg_cram_decode_bam_host")
        return out[:total.value], rec_off, st

    def cram_batch_stage(self, slice_array, nslices, major, nref, total_bases):
        """hg_cram_batch_stage: the slices' blocks, reference spans and header tables go to HBM once -> opaque batch handle"""
        h = _vp()
        check(lib.hg_cram_batch_stage(self._h, nslices, C.cast(slice_array, _vp), major, nref, total_bases, C.byref(h)), "hg_cram_batch_stage")
        return h

    def cram_batch_decode_bam(self, batch, nslices, rg_names=(), name_prefix=None):
        """hg_cram_batch_decode_bam_dev: one decoding run, BAM stream left on the device -> (device pointer, bytes, records, slices decoded by the passes, status)"""
        import numpy as np
        rg = [r.encode() if isinstance(r, str) else r for r in rg_names]
        rgp = (C.c_char_p * max(len(rg), 1))(*rg) if rg else None
        d = _vp(); nb, nr, nf = C.c_uint64(), C.c_uint64(), C.c_uint64(); st = np.full(nslices, 9, np.int32)
        rc = lib.hg_cram_batch_decode_bam_dev(self._h, batch, C.cast(rgp, _vp) if rg else None, len(rg), name_prefix, C.byref(d), C.byref(nb), C.byref(nr), C.byref(nf), st.ctypes.data)
        if rc not in (0, -6):
            check(rc, "hg_cram_batch_decode_bam_dev")
        return d.value, nb.value, nr.value, nf.value, st

    def cram_batch_read_bam(self, batch, nbytes):
        import numpy as np
        out = np.empty(max(nbytes, 1), np.uint8)
        check(lib.hg_cram_batch_read_bam(self._h, batch, out.ctypes.data, nbytes), "hg_cram_batch_read_bam")
        return out[:nbytes]

    def cram_batch_free(self, batch):
        lib.hg_cram_batch_free(self._h, batch)

    def tok3_encode_host(self, datas, use_arith):
        """Tokenise + entropy-code each buffer of NUL-terminated names -> list of CRAM method-8 payloads (b"" = not names)."""
        import numpy as np
        if not datas:
            return []
        ins, outs, ip, op, il, ol = self._ptr_batch(datas, lib.hg_tok3_compress_bound)
        ua = np.array(use_arith, dtype=np.uint8)
        check(lib.hg_tok3_encode_host(self._h, ip, il.ctypes.data, ua.ctypes.data, len(datas), op, ol.ctypes.data), "hg_tok3_encode_host")
        return [outs[i].raw[:int(ol[i])] for i in range(len(datas))]

    def ransnx16_encode_host(self, datas, flags):
        """rANS Nx16-encode each bytes object in `datas` with the matching flag byte -> list of streams."""
        import numpy as np
        n = len(datas)
        if n == 0:
            return []
        ins = [(C.c_char * max(len(d), 1)).from_buffer_copy(d if len(d) else b"\0") for d in datas]
        outs = [C.create_string_buffer(lib.hg_ransnx16_compress_bound(len(d))) for d in datas]
        in_ptr = (_vp * n)(*[C.addressof(x) for x in ins])
        out_ptr = (_vp * n)(*[C.addressof(x) for x in outs])
        in_len = np.array([len(d) for d in datas], dtype=np.uint32)
        fl = np.array(flags, dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.uint32)
        check(lib.hg_ransnx16_encode_host(self._h, in_ptr, in_len.ctypes.data, fl.ctypes.data, n, out_ptr,
                                          out_len.ctypes.data), "hg_ransnx16_encode_host")
        return [outs[i].raw[:int(out_len[i])] for i in range(n)]

    # -- device-resident entry points (torch tensors or raw pointers) ----------
    def bgzf_inflate_dev(self, d_comp: int, comp_len: int, d_desc: int, nblocks: int, d_out: int,
                         out_cap: int, d_status: int, stream: int = 0):
        check(lib.hg_bgzf_inflate_dev(self._h, d_comp, comp_len, d_desc, nblocks, d_out, out_cap,
                                      d_status, stream), "hg_bgzf_inflate_dev")

    def crc32_dev(self, d_data: int, d_off: int, d_len: int, n: int, d_crc: int, stream: int = 0):
        check(lib.hg_crc32_dev(self._h, d_data, d_off, d_len, n, d_crc, stream), "hg_crc32_dev")
