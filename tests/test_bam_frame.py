"""BAM record framing: bam_hdr_read's walk, bam_read1's framing + sanity checks, nibble2base.
CPU part pins oracle/bam_oracle.c against artefacts written by reference htslib: the .bai indexes of its own BAM
fixtures hold record boundaries (chunk begins, linear-index offsets) and per-reference mapped / unmapped counts.
GPU part: the device framing is identical to the oracle on fixtures, synthetic BAM, corrupted and truncated streams."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

from htslib_amd import synth
from tests import refutil

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BAMS = ["colons.bam", "range.bam", "mpileup__small.bam", "no_hdr_sq_1.bam", "bgzf_boundaries__bgzf_boundaries1.bam",
        "bgzf_boundaries__bgzf_boundaries2.bam", "bgzf_boundaries__bgzf_boundaries3.bam"]


class BamOracle:
    def __init__(self):
        L = C.CDLL(os.path.join(refutil.ROOT, "oracle", "liboracle.so"))
        L.orc_bam_header.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        L.orc_bam_frame.restype = C.c_long
        L.orc_bam_frame.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_long, C.POINTER(C.c_uint64)]
        L.orc_nibble2base.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        self.L = L

    def header(self, b):
        n, f = C.c_int32(), C.c_uint64()
        rc = self.L.orc_bam_header(b, len(b), C.byref(n), C.byref(f))
        return rc, n.value, f.value

    def frame(self, b, first):
        bad = C.c_uint64()
        n = self.L.orc_bam_frame(b, len(b), first, None, 0, C.byref(bad))
        if n < 0:
            return n, bad.value, None
        off = np.zeros(max(n, 1), dtype=np.uint64)
        self.L.orc_bam_frame(b, len(b), first, off.ctypes.data, n, C.byref(bad))
        return n, None, off[:n]

    def bases(self, b, rec):
        x = rec + 4
        l_qname, n_cigar, l_qseq = b[x + 8], struct.unpack_from("<H", b, x + 12)[0], struct.unpack_from("<i", b, x + 16)[0]
        out = C.create_string_buffer(max(l_qseq, 1))
        self.L.orc_nibble2base(b[x + 32 + l_qname + 4 * n_cigar:], out, l_qseq)
        return out.raw[:l_qseq]


@pytest.fixture(scope="module")
def borc(built):
    return BamOracle()


def plain_of(name):
    return open(os.path.join(GOLD, "bgzf", name + ".plain"), "rb").read()


def voffset_map(bgzf):
    """compressed block offset -> uncompressed stream offset"""
    m, p, u = {}, 0, 0
    while p + 18 <= len(bgzf):
        bsize = struct.unpack_from("<H", bgzf, p + 16)[0] + 1
        m[p] = u
        u += struct.unpack_from("<I", bgzf, p + bsize - 4)[0]
        p += bsize
    return m


def parse_bai(b):
    assert b[:4] == b"BAI\x01"
    n_ref = struct.unpack_from("<i", b, 4)[0]
    p = 8
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", b, p)[0]; p += 4
        starts, meta = [], None
        for _ in range(n_bin):
            bin_, n_chunk = struct.unpack_from("<Ii", b, p); p += 8
            chunks = [struct.unpack_from("<QQ", b, p + 16 * k) for k in range(n_chunk)]; p += 16 * n_chunk
            if bin_ == 37450:
                meta = chunks
            else:
                starts += [c[0] for c in chunks]
        n_intv = struct.unpack_from("<i", b, p)[0]; p += 4
        starts += [v for v in struct.unpack_from("<%dQ" % n_intv, b, p) if v]; p += 8 * n_intv
        refs.append((starts, meta))
    return refs


@pytest.mark.parametrize("name", ["colons.bam", "range.bam"])
def test_oracle_framing_matches_the_index_reference_htslib_wrote(borc, name):
    bgzf = open(os.path.join(GOLD, "bgzf", name), "rb").read()
    plain = plain_of(name)
    rc, n_ref, first = borc.header(plain)
    assert rc == 0
    n, _, off = borc.frame(plain, first)
    assert n > 0 and off[0] == first
    starts = set(int(o) for o in off)
    vmap = voffset_map(bgzf)
    refs = parse_bai(open(os.path.join(GOLD, "bam", name + ".bai"), "rb").read())
    assert len(refs) == n_ref
    tid = np.array([struct.unpack_from("<i", plain, int(o) + 4)[0] for o in off])
    flag = np.array([struct.unpack_from("<I", plain, int(o) + 4 + 12)[0] >> 16 for o in off])
    checked = 0
    for r, (vstarts, meta) in enumerate(refs):
        for v in vstarts:                                   # every chunk begin / linear-index entry is a record boundary
            assert vmap[v >> 16] + (v & 0xffff) in starts
            checked += 1
        if meta:                                            # pseudo-bin: (ref_beg, ref_end), (n_mapped, n_unmapped)
            assert meta[1][0] == int(((tid == r) & ((flag & 4) == 0)).sum())
            assert meta[1][1] == int(((tid == r) & ((flag & 4) != 0)).sum())
    assert checked > 0


def test_oracle_on_all_fixtures_and_error_codes(borc):
    for name in BAMS:
        plain = plain_of(name)
        rc, n_ref, first = borc.header(plain)
        assert rc == 0, name
        n, bad, off = borc.frame(plain, first)
        assert n >= 0, name
        if n:
            assert borc.frame(plain[:-3], first)[0] == -2                       # truncated last record
            assert len(borc.bases(plain, int(off[0]))) == struct.unpack_from("<i", plain, int(off[0]) + 20)[0]
    assert borc.header(b"BAM\x02" + bytes(20))[0] == -1
    assert borc.L.orc_nibble2base is not None
    out = C.create_string_buffer(5)
    borc.L.orc_nibble2base(bytes([0x12, 0x48, 0xf0]), out, 5)
    assert out.raw == b"ACGTN"


def corrupt_cases(plain, first, off):
    bad_bl = bytearray(plain); struct.pack_into("<i", bad_bl, int(off[len(off) // 2]), 16)             # block_len < 32
    bad_q = bytearray(plain); bad_q[int(off[len(off) // 3]) + 4 + 8] = 0                                  # l_qname 0
    bad_seq = bytearray(plain); struct.pack_into("<i", bad_seq, int(off[5]) + 4 + 16, 1 << 28)           # l_qseq does not fit
    return [bytes(bad_bl), bytes(bad_q), bytes(bad_seq), plain[:int(off[-1]) + 20], plain[:int(off[-1]) + 2]]


def giant_record(l_seq=200_000):
    name = b"giant\0"
    core = struct.pack("<iiIIiiii", 0, 100, (4680 << 16) | (60 << 8) | len(name), (0 << 16) | 1, l_seq, -1, -1, 0)
    body = core + name + struct.pack("<I", (l_seq << 4) | 0) + bytes([0x12]) * ((l_seq + 1) // 2) + bytes([30]) * l_seq
    return struct.pack("<i", len(body)) + body


@pytest.mark.gpu
def test_gpu_framing_and_bases_match_oracle(engine, borc):
    import torch
    from htslib_amd import _native as nat
    streams = [plain_of(n) for n in BAMS]
    big = synth.bam_stream(12 << 20)[0]
    hdr_len = borc.header(big)[2]
    streams.append(big)
    streams.append(big[:hdr_len] + big[hdr_len:hdr_len + 300_000] + giant_record() + big[hdr_len + 300_000:hdr_len + 900_000] + giant_record(70_000))
    n0, _, off0 = borc.frame(big, hdr_len)
    streams += corrupt_cases(big, hdr_len, off0)
    for b in streams:
        rc, n_ref, first = borc.header(b)
        nref_g, first_g = C.c_int32(), C.c_uint64()
        assert nat.lib.hg_bam_header_host(b, len(b), C.byref(nref_g), C.byref(first_g)) == 0
        assert (nref_g.value, first_g.value) == (n_ref, first)
        want_n, want_bad, want_off = borc.frame(b, first)
        d = torch.frombuffer(bytearray(b + bytes(64)), dtype=torch.uint8).cuda()
        bad = C.c_uint64(0)
        n = nat.lib.hg_bam_frame_dev(engine._h, d.data_ptr(), len(b), first, n_ref, None, 0, C.byref(bad), None)
        assert n == want_n, (len(b), n, want_n)
        if want_n < 0:
            assert bad.value == want_bad
            continue
        d_off = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
        assert nat.lib.hg_bam_frame_dev(engine._h, d.data_ptr(), len(b), first, n_ref, d_off.data_ptr(), n, C.byref(bad), None) == n
        got = d_off.cpu().numpy().astype(np.uint64)[:n]
        assert (got == want_off).all()
        # bases
        d_boff = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        tot = C.c_uint64(0)
        assert nat.lib.hg_bam_bases_dev(engine._h, d.data_ptr(), d_off.data_ptr(), n, d_boff.data_ptr(), None, 0, C.byref(tot), None) == 0
        d_bases = torch.zeros(tot.value + 64, dtype=torch.uint8, device="cuda")
        assert nat.lib.hg_bam_bases_dev(engine._h, d.data_ptr(), d_off.data_ptr(), n, d_boff.data_ptr(), d_bases.data_ptr(), tot.value, C.byref(tot), None) == 0
        boff = d_boff.cpu().numpy()
        bases = d_bases.cpu().numpy().tobytes()
        d_quals = torch.zeros(tot.value + 64, dtype=torch.uint8, device="cuda")
        assert nat.lib.hg_bam_quals_dev(engine._h, d.data_ptr(), d_off.data_ptr(), n, d_boff.data_ptr(), d_quals.data_ptr(), None) == 0
        quals = d_quals.cpu().numpy().tobytes()
        cols = {k: torch.zeros(max(n, 1), dtype=t, device="cuda") for k, t in (("tid", torch.int32), ("pos", torch.int32), ("bin", torch.int16),
                ("mapq", torch.uint8), ("l_qname", torch.uint8), ("flag", torch.int16), ("n_cigar", torch.int16), ("l_qseq", torch.int32),
                ("mtid", torch.int32), ("mpos", torch.int32), ("isize", torch.int32))}
        cc = nat.BamCoreCols(**{k: v.data_ptr() for k, v in cols.items()})
        assert nat.lib.hg_bam_core_dev(engine._h, d.data_ptr(), d_off.data_ptr(), n, C.byref(cc), None) == 0
        torch.cuda.synchronize()
        host = {k: v.cpu().numpy() for k, v in cols.items()}
        for i in list(range(min(n, 40))) + list(range(max(0, n - 40), n)) + list(range(0, n, max(1, n // 200))):
            r = int(want_off[i])
            assert bases[boff[i]:boff[i + 1]] == borc.bases(b, r)
            tid, pos, x2, x3, l_qseq, mtid, mpos, isize = struct.unpack_from("<iiIIiiii", b, r + 4)
            assert (host["tid"][i], host["pos"][i], host["l_qseq"][i], host["mtid"][i], host["mpos"][i], host["isize"][i]) == (tid, pos, l_qseq, mtid, mpos, isize)
            assert (int(host["bin"][i]) & 0xffff, host["mapq"][i], host["l_qname"][i], int(host["flag"][i]) & 0xffff, int(host["n_cigar"][i]) & 0xffff) == \
                   (x2 >> 16, (x2 >> 8) & 0xff, x2 & 0xff, x3 >> 16, x3 & 0xffff)
            qs = r + 36 + (x2 & 0xff) + 4 * (x3 & 0xffff) + (l_qseq + 1) // 2
            assert quals[boff[i]:boff[i + 1]] == bytes(0xff if v == 0xff else v + 33 for v in b[qs:qs + l_qseq])
