"""BAI construction: what `samtools index` computes -- hts_idx_push / hts_idx_finish /
compress_binning / idx_save_core.  The oracle (oracle/bam_oracle.c orc_bai_build) is pinned against the .bai files
that reference htslib wrote for its own BAM fixtures: byte-identical after sorting the bins of each reference (htslib
writes them in hash-table order).  GPU part: hg_bai_build_dev == oracle."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from htslib_amd import synth
from tests import refutil
from tests.test_bam_frame import GOLD, BamOracle, plain_of


def block_table(bgzf):
    rows, p, u = [], 0, 0
    while p + 18 <= len(bgzf):
        bsize = struct.unpack_from("<H", bgzf, p + 16)[0] + 1
        isize = struct.unpack_from("<I", bgzf, p + bsize - 4)[0]
        rows.append((p, u, isize, 0))
        u += isize
        p += bsize
    return np.array(rows, dtype=[("coff", "<u8"), ("uoff", "<u8"), ("ulen", "<u4"), ("pad", "<u4")]), p


def canonical_bai(b):
    """.bai bytes with the bins of every reference in ascending order."""
    assert b[:4] == b"BAI\x01"
    n_ref = struct.unpack_from("<i", b, 4)[0]
    p, out = 8, [b[:8]]
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", b, p)[0]; p += 4
        bins = []
        for _ in range(n_bin):
            bin_, n_chunk = struct.unpack_from("<Ii", b, p)
            bins.append((bin_, b[p:p + 8 + 16 * n_chunk])); p += 8 + 16 * n_chunk
        n_intv = struct.unpack_from("<i", b, p)[0]
        lin = b[p:p + 4 + 8 * n_intv]; p += 4 + 8 * n_intv
        out.append(struct.pack("<i", n_bin) + b"".join(x for _, x in sorted(bins)) + lin)
    out.append(b[p:])
    return b"".join(out)


class BaiOracle(BamOracle):
    def __init__(self):
        super().__init__()
        self.L.orc_bai_build.restype = C.c_long
        self.L.orc_bai_build.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p, C.c_long, C.c_uint64, C.c_char_p, C.c_long]

    def idx(self, plain, bgzf, csi, min_shift, n_lvls):
        self.L.orc_idx_build.restype = C.c_long
        self.L.orc_idx_build.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p, C.c_long, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                         C.c_char_p, C.c_long]
        rc, n_ref, first = self.header(plain)
        assert rc == 0
        blk, fsize = block_table(bgzf)
        out = C.create_string_buffer(1 << 22)
        n = self.L.orc_idx_build(plain, len(plain), first, n_ref, blk.ctypes.data, len(blk), fsize, csi, min_shift, n_lvls, out, len(out))
        return out.raw[:n] if n >= 0 else None

    def bai(self, plain, bgzf):
        rc, n_ref, first = self.header(plain)
        assert rc == 0
        blk, fsize = block_table(bgzf)
        out = C.create_string_buffer(1 << 22)
        n = self.L.orc_bai_build(plain, len(plain), first, n_ref, blk.ctypes.data, len(blk), fsize, out, len(out))
        return out.raw[:n] if n >= 0 else None


@pytest.fixture(scope="module")
def iorc(built):
    return BaiOracle()


@pytest.mark.parametrize("name", ["colons.bam", "range.bam"])
def test_oracle_reproduces_the_index_reference_htslib_wrote(iorc, name):
    bgzf = open(os.path.join(GOLD, "bgzf", name), "rb").read()
    want = open(os.path.join(GOLD, "bam", name + ".bai"), "rb").read()
    got = iorc.bai(plain_of(name), bgzf)
    assert got is not None
    assert got == canonical_bai(want)


def ref_lengths(plain):
    n_ref = struct.unpack_from("<i", plain, 8 + struct.unpack_from("<I", plain, 4)[0])[0]
    p = 8 + struct.unpack_from("<I", plain, 4)[0] + 4
    out = []
    for _ in range(n_ref):
        l = struct.unpack_from("<i", plain, p)[0]
        out.append(struct.unpack_from("<I", plain, p + 4 + l)[0]); p += 8 + l
    return out


def test_oracle_reproduces_the_csi_reference_htslib_wrote(iorc):
    """`samtools index -c`: the CSI layout; depth from hts_adjust_csi_settings (14 / 2 for a 1 Mbp reference)."""
    from htslib_amd import _native as nat
    name = "no_hdr_sq_1.bam"
    plain = plain_of(name)
    want = open(os.path.join(GOLD, "bgzf", name + ".csi.plain"), "rb").read()
    depth = nat.lib.hg_csi_levels(max(ref_lengths(plain)), 14)
    assert depth == struct.unpack_from("<i", want, 8)[0] == 2
    assert nat.lib.hg_csi_levels(249_250_621, 14) == 5 and nat.lib.hg_csi_levels(600_000_000, 14) == 6
    assert iorc.idx(plain, open(os.path.join(GOLD, "bgzf", name), "rb").read(), 1, 14, depth) == want


def test_oracle_rejects_unsorted_input(iorc):
    plain, bgzf = synth.bam_bgzf(1 << 20)
    assert iorc.bai(plain, bgzf) is not None
    rc, n_ref, first = iorc.header(plain)
    n, _, off = iorc.frame(plain, first)
    a, b = int(off[10]), int(off[11])
    swapped = plain[:a] + plain[b:int(off[12])] + plain[a:b] + plain[int(off[12]):]     # positions now go backwards (or stay equal)
    pa = struct.unpack_from("<i", plain, a + 8)[0]; pb = struct.unpack_from("<i", plain, b + 8)[0]
    if pa != pb:
        assert iorc.bai(swapped, bgzf) is None


def gpu_bai(engine, plain, bgzf, iorc, csi=0, min_shift=14, n_lvls=5):
    import torch
    from htslib_amd import _native as nat
    rc, n_ref, first = iorc.header(plain)
    # reference lengths from the header (l_ref follows each name)
    p = 8 + struct.unpack_from("<I", plain, 4)[0] + 4
    ref_len = []
    for _ in range(n_ref):
        l = struct.unpack_from("<i", plain, p)[0]
        ref_len.append(struct.unpack_from("<I", plain, p + 4 + l)[0]); p += 8 + l
    rl = np.array(ref_len + [0], dtype=np.uint32)
    d = torch.frombuffer(bytearray(plain + bytes(64)), dtype=torch.uint8).cuda()
    bad = C.c_uint64(0)
    n = nat.lib.hg_bam_frame_dev(engine._h, d.data_ptr(), len(plain), first, n_ref, None, 0, C.byref(bad), None)
    assert n >= 0
    d_off = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
    assert nat.lib.hg_bam_frame_dev(engine._h, d.data_ptr(), len(plain), first, n_ref, d_off.data_ptr(), n, C.byref(bad), None) == n
    blk, fsize = block_table(bgzf)
    desc = np.zeros(len(blk), dtype=nat.DESC_DTYPE) if hasattr(nat, "DESC_DTYPE") else None
    if desc is None:
        desc = np.zeros(len(blk), dtype=[("coff", "<u8"), ("uoff", "<u8"), ("clen", "<u4"), ("ulen", "<u4")])
    desc["coff"], desc["uoff"], desc["ulen"] = blk["coff"], blk["uoff"], blk["ulen"]
    out = C.create_string_buffer(1 << 24)
    r = nat.lib.hg_idx_build_dev(engine._h, d.data_ptr(), len(plain), first, n_ref, rl.ctypes.data, d_off.data_ptr(), n, desc.ctypes.data,
                                 len(desc), fsize, csi, min_shift, n_lvls, out, len(out), None)
    return r, out.raw[:max(r, 0)]


@pytest.mark.gpu
def test_gpu_index_equals_oracle_and_reference(engine, iorc):
    for name in ("colons.bam", "range.bam", "mpileup__small.bam", "bgzf_boundaries__bgzf_boundaries1.bam", "bgzf_boundaries__bgzf_boundaries3.bam"):
        bgzf = open(os.path.join(GOLD, "bgzf", name), "rb").read()
        plain = plain_of(name)
        want = iorc.bai(plain, bgzf)
        r, got = gpu_bai(engine, plain, bgzf, iorc)
        if want is None:
            assert r == -5, name
        else:
            assert r == len(want) and got == want, name
    for name in ("colons.bam", "range.bam"):                               # and through the oracle to reference htslib's own files
        ref = open(os.path.join(GOLD, "bam", name + ".bai"), "rb").read()
        assert gpu_bai(engine, plain_of(name), open(os.path.join(GOLD, "bgzf", name), "rb").read(), iorc)[1] == canonical_bai(ref)
    plain, bgzf = synth.bam_bgzf(24 << 20)                                # ~80 k records, many bins, several references
    want = iorc.bai(plain, bgzf)
    assert want is not None and len(want) > 1000
    r, got = gpu_bai(engine, plain, bgzf, iorc)
    assert r == len(want) and got == want
    # CSI: the reference's own fixture, and other depths / shifts against the oracle
    name = "no_hdr_sq_1.bam"
    bg = open(os.path.join(GOLD, "bgzf", name), "rb").read()
    assert gpu_bai(engine, plain_of(name), bg, iorc, 1, 14, 2)[1] == open(os.path.join(GOLD, "bgzf", name + ".csi.plain"), "rb").read()
    for ms, nl in ((14, 5), (12, 6), (16, 4)):
        want = iorc.idx(plain, bgzf, 1, ms, nl)
        assert want is not None and gpu_bai(engine, plain, bgzf, iorc, 1, ms, nl)[1] == want
    # unsorted input is refused like `samtools index` does
    rc, n_ref, first = iorc.header(plain)
    n, _, off = iorc.frame(plain, first)
    k = next(i for i in range(10, n - 2) if struct.unpack_from("<i", plain, int(off[i]) + 8)[0] < struct.unpack_from("<i", plain, int(off[i + 1]) + 8)[0]
             and struct.unpack_from("<i", plain, int(off[i]) + 4)[0] == struct.unpack_from("<i", plain, int(off[i + 1]) + 4)[0])
    a, b, c = int(off[k]), int(off[k + 1]), int(off[k + 2])
    swapped = plain[:a] + plain[b:c] + plain[a:b] + plain[c:]
    assert iorc.bai(swapped, bgzf) is None and gpu_bai(engine, swapped, bgzf, iorc)[0] == -5
