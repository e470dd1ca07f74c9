"""rANS Nx16 (CRAM 3.1 block method 5) -- PARITY UNPINNED (see oracle/ransnx16_oracle.c): the
reference holds neither the codec source (htscodecs submodule absent) nor any Nx16 stream.  CPU part:
the oracle's encoder/decoder are mutually consistent over every flag combination; GPU part: the
gfx950 decoder is bit-exact with the oracle for the flag sets it supports."""
import numpy as np
import pytest

from tests import refutil
from tests.test_rans4x8 import synth_series

ALL_FLAGS = [0, 1, 4, 5, 0x20, 0x80, 0x81, 0x84, 0x40, 0x41, 0x44, 0xC0, 0xC1, 0xC5, 0x08, 0x09, 0x0C, 0x0D]
GPU_FLAGS = [0, 1, 4, 5, 0x20, 0x24]
SIZES = (0, 1, 2, 3, 4, 7, 8, 9, 31, 32, 33, 63, 64, 65, 100, 1000, 4097, 150_000)


@pytest.fixture(scope="module")
def norc(built):
    return refutil.RansNx16Oracle()


@pytest.mark.parametrize("kind", ["qual4", "qual41", "bases", "bytes", "const"])
def test_oracle_roundtrip_all_flags(norc, kind):
    rng = np.random.default_rng(len(kind))
    for n in SIZES:
        d = synth_series(rng, kind, n)
        for fl in ALL_FLAGS:
            e = norc.encode(d, fl)
            rc, out = norc.decode(e, len(d))
            assert rc == 0 and out == d, (kind, n, hex(fl))
            if n >= 1000 and not (fl & 0x20):
                assert norc.decode(e[:len(e) // 2], len(d))[0] == -1     # truncation is detected


def test_oracle_order1_and_transforms_actually_compress(norc):
    rng = np.random.default_rng(1)
    q = synth_series(rng, "qual4", 300_000)
    o0, o1, o1p = len(norc.encode(q, 0)), len(norc.encode(q, 1)), len(norc.encode(q, 0x81))
    assert o1 < 0.8 * o0 and o1p < 0.5 * len(q)
    assert len(norc.encode(bytes(100_000), 0x40)) < 200               # RLE collapses runs
    assert abs(len(norc.encode(q, 4)) - o0) < 200                      # 32-way == 4-way + 28 more states


@pytest.mark.gpu
def test_gpu_matches_oracle(engine, norc):
    rng = np.random.default_rng(42)
    blocks, want = [], []
    for kind in ("qual4", "qual41", "bases", "bytes", "const"):
        for n in SIZES:
            d = synth_series(rng, kind, n)
            for fl in GPU_FLAGS:
                blocks.append((5, norc.encode(d, fl), len(d))); want.append(d)
    outs, st = engine.cram_uncompress_blocks(blocks)
    assert (st == 0).all()
    assert outs == want
    # flags the kernel does not handle yet are reported, not mis-decoded
    d = synth_series(rng, "qual4", 5000)
    outs, st = engine.cram_uncompress_blocks([(5, norc.encode(d, fl), len(d)) for fl in (0x80, 0x40, 0x08)])
    assert list(st) == [-3, -3, -3]


@pytest.mark.gpu
def test_gpu_big_quality_streams_and_fuzz(engine, norc):
    rng = np.random.default_rng(8)
    d = synth_series(rng, "qual41", 1_500_000)                         # one slice worth of QS
    streams = [norc.encode(d, fl) for fl in (0, 1, 4, 5)]
    outs, st = engine.cram_uncompress_blocks([(5, s, len(d)) for s in streams])
    assert (st == 0).all() and all(o == d for o in outs)
    small = synth_series(rng, "qual41", 20_000)
    base = [norc.encode(small, fl) for fl in (0, 1, 4, 5)]
    bad = []
    for rep in range(200):
        b = bytearray(base[rep & 3])
        pos = int(rng.integers(1, len(b)))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(b))
    bad += [base[0][:-5], base[1][:40], b"", b"\x05"]
    outs, st = engine.cram_uncompress_blocks([(5, b, len(small)) for b in bad])
    for b, o, s in zip(bad, outs, st):
        rc, want = norc.decode(b, len(small))
        if rc == 0 and len(want) == len(small):
            assert s == 0 and o == want
        else:
            assert s != 0


@pytest.mark.gpu
def test_gpu_encoder_is_byte_identical_to_oracle_and_decodes(engine, norc):
    """The gfx950 encoder reproduces the oracle's stream byte for byte (same normalisation, table
    layout and word order) and its output decodes on the GPU and in the oracle."""
    rng = np.random.default_rng(21)
    datas, flags = [], []
    for kind in ("qual4", "qual41", "bases", "bytes", "const"):
        for n in SIZES:
            d = synth_series(rng, kind, n)
            for fl in (0, 1, 4, 5, 0x20, 0x10, 0x11, 0x15):
                datas.append(d); flags.append(fl)
    enc = engine.ransnx16_encode_host(datas, flags)
    for d, fl, e in zip(datas, flags, enc):
        assert e == norc.encode(d, fl), (len(d), hex(fl))
    sized = [(5, e, len(d)) for d, fl, e in zip(datas, flags, enc)]
    outs, st = engine.cram_uncompress_blocks(sized)
    assert (st == 0).all() and outs == datas
