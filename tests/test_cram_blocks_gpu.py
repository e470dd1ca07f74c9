"""GPU tests of the CRAM block layer (hg_cram_uncompress_blocks_host = batch cram_uncompress_block)
on every block of the reference's CRAM v3.0 fixtures + synthetic GZIP/rANS/RAW mixes."""
import gzip
import json
import os
import zlib

import numpy as np
import pytest

from tests import refutil
from htslib_amd import synth

pytestmark = pytest.mark.gpu
VEC = json.load(open(os.path.join(refutil.ROOT, "tests", "golden", "cram_blocks.json")))


def test_reference_cram_fixture_blocks(engine):
    blocks = [(v["method"], bytes.fromhex(v["data_hex"]), v["usize"]) for v in VEC]
    outs, st = engine.cram_uncompress_blocks(blocks)
    assert {v["method"] for v in VEC} == {0, 1, 4}
    assert (st == 0).all()
    rorc = refutil.Rans4x8Oracle()
    checked = 0
    for v, o in zip(VEC, outs):
        assert len(o) == v["usize"]
        if v["expected_hex"] is not None:
            assert o.hex() == v["expected_hex"]; checked += 1
        elif v["method"] == 4 and v["usize"]:
            assert o == rorc.decode(bytes.fromhex(v["data_hex"]))[1]
    assert checked >= 149


def test_large_gzip_members_and_mixed_batch(engine):
    """CRAM GZIP blocks are whole data series (MBs), not 64 KiB: 32 KiB window across many deflate
    blocks, gzip header variants, CRC/ISIZE trailer."""
    rng = np.random.default_rng(9)
    rorc = refutil.Rans4x8Oracle()
    plain_bam, _ = synth.bam_bgzf(3 << 20)
    series = [plain_bam, synth.fastq(1_500_000), bytes(2_000_000), rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(), b"x"]
    blocks, want = [], []
    for d in series:
        for lvl in (1, 6, 9):
            co = zlib.compressobj(lvl, zlib.DEFLATED, 15 + 16, 9)       # gzip wrapper, memLevel 9 like zlib_mem_deflate
            blocks.append((1, co.compress(d) + co.flush(), len(d))); want.append(d)
        blocks.append((1, gzip.compress(d, 6, mtime=12345), len(d))); want.append(d)
        blocks.append((0, d, len(d))); want.append(d)
        blocks.append((4, rorc.encode(d[:200_000], 1), len(d[:200_000]))); want.append(d[:200_000])
    # a member with FNAME + FCOMMENT + FEXTRA header fields
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(series[1]) + raw.flush()
    import struct
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16]) + b"\0\0\0\0\0\xff" + struct.pack("<H", 5) + b"ABCDE" + b"name\0" + b"comment\0"
    blocks.append((1, hdr + body + struct.pack("<II", zlib.crc32(series[1]), len(series[1])), len(series[1]))); want.append(series[1])
    outs, st = engine.cram_uncompress_blocks(blocks)
    assert (st == 0).all()
    assert outs == want


def test_status_codes(engine):
    d = synth.fastq(100_000)
    good = gzip.compress(d, 6)
    bad_crc = good[:-8] + bytes([good[-8] ^ 1]) + good[-7:]
    blocks = [(1, good, len(d)), (1, bad_crc, len(d)), (1, good, len(d) - 1), (1, good[:len(good) // 2], len(d)),
              (6, b"\x00" * 20, 10), (2, b"BZh", 10), (0, d, len(d)), (0, d, len(d) + 1), (1, b"", 0)]
    outs, st = engine.cram_uncompress_blocks(blocks)
    assert list(st) == [0, -2, -1, -1, -3, -3, 0, -1, 0]
    assert outs[0] == d and outs[6] == d and outs[1] is None
