"""rANS Nx16 (CRAM 3.1 block method 5) -- PARITY UNPINNED (see oracle/ransnx16_oracle.c): the
reference holds neither the codec source (htscodecs submodule absent) nor any Nx16 stream.  CPU part:
the oracle's encoder/decoder are mutually consistent over every flag combination; GPU part: the
gfx950 decoder is bit-exact with the oracle for the flag sets it supports."""
import numpy as np
import pytest

from tests import refutil
from tests.test_rans4x8 import synth_series

ALL_FLAGS = [0, 1, 4, 5, 0x20, 0x80, 0x81, 0x84, 0x40, 0x41, 0x44, 0xC0, 0xC1, 0xC5, 0x08, 0x09, 0x0C, 0x0D,
             0x24, 0xA0, 0x60, 0xE0, 0xE4]
GPU_FLAGS = ALL_FLAGS
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


def runs_series(rng, n, nsym=6, mean=40):
    """Long runs of a few symbols: what the RLE transform is for (and big enough run counts that the
    oracle rANS-codes the RLE meta stream)."""
    out = bytearray()
    while len(out) < n:
        out += bytes([int(rng.integers(0, nsym)) + 33]) * int(rng.geometric(1.0 / mean))
    return bytes(out[:n])


@pytest.mark.gpu
def test_gpu_transforms_rle_pack_stripe(engine, norc):
    """PACK / RLE / STRIPE are undone on the device (ransnx16_xform.hip) after the entropy decode."""
    rng = np.random.default_rng(77)
    cases = []
    for n in (1, 5, 63, 64, 65, 4096, 70_000, 1_200_000):
        cases += [runs_series(rng, n), runs_series(rng, n, nsym=2, mean=3), runs_series(rng, n, nsym=16, mean=900),
                  bytes([65]) * n, synth_series(rng, "bases", n)]
    u32 = rng.integers(0, 50_000, 100_000, dtype=np.uint32).tobytes()          # what STRIPE is for
    cases += [u32, u32[:-3], rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()]
    blocks, want = [], []
    for d in cases:
        for fl in (0x40, 0x41, 0x80, 0xC0, 0xC1, 0xC5, 0x08, 0x09, 0x0D, 0xE0, 0x50, 0x90):
            e = norc.encode(d, fl)
            if not fl & 0x10:                    # NOSZ streams need the size from outside (the CRAM block header)
                rc, back = norc.decode(e, len(d))
                assert rc == 0 and back == d
            blocks.append((5, e, len(d))); want.append(d)
    assert any(not (e[1][0] & 0x10) and (e[1][0] & 0x40) for e in blocks)
    outs, st = engine.cram_uncompress_blocks(blocks)
    bad = [(i, hex(blocks[i][1][0]), blocks[i][2]) for i in range(len(blocks)) if st[i] != 0 or outs[i] != want[i]]
    assert not bad, bad[:10]


@pytest.mark.gpu
def test_gpu_big_quality_streams_and_fuzz(engine, norc):
    rng = np.random.default_rng(8)
    d = synth_series(rng, "qual41", 1_500_000)                         # one slice worth of QS
    streams = [norc.encode(d, fl) for fl in (0, 1, 4, 5)]
    outs, st = engine.cram_uncompress_blocks([(5, s, len(d)) for s in streams])
    assert (st == 0).all() and all(o == d for o in outs)
    small = synth_series(rng, "qual41", 20_000)
    runs = (small[:200] + bytes([70]) * 300) * 40
    base = [norc.encode(small, fl) for fl in (0, 1, 4, 5)] + [norc.encode(runs, fl) for fl in (0xC1, 0x40, 0x09, 0x80)]
    assert len(runs) == len(small)
    bad = []
    for rep in range(400):
        b = bytearray(base[rep & 7])
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


@pytest.mark.gpu
def test_gpu_encoder_transforms_match_oracle(engine, norc):
    """The RANS_PR* flag sets htslib asks for ({1,64,9,128,129,192,193}, cram_io.c:1856) plus the other
    PACK / RLE / STRIPE / CAT / X32 combinations: GPU-encoded stream == oracle stream, and it decodes."""
    rng = np.random.default_rng(5)
    datas, flags = [], []
    series = []
    for n in (0, 1, 3, 4, 7, 8, 9, 63, 64, 65, 127, 128, 129, 200, 4096, 70_000, 400_000):
        series += [runs_series(rng, n), runs_series(rng, n, nsym=2, mean=3), runs_series(rng, n, nsym=16, mean=900),
                   bytes([65]) * n, synth_series(rng, "bases", n), synth_series(rng, "qual41", n)]
    series.append(rng.integers(0, 50_000, 50_000, dtype=np.uint32).tobytes())
    for d in series:
        for fl in (1, 64, 9, 128, 129, 192, 193, 0x08, 0x0C, 0x0D, 0x41, 0x44, 0xC5, 0xA0, 0x60, 0xE0, 0xE4, 0x50, 0x90, 0xD1, 0x18, 0x19):
            datas.append(d); flags.append(fl)
    enc = engine.ransnx16_encode_host(datas, flags)
    bad = [(len(d), hex(fl)) for d, fl, e in zip(datas, flags, enc) if e != norc.encode(d, fl)]
    assert not bad, bad[:12]
    outs, st = engine.cram_uncompress_blocks([(5, e, len(d)) for d, e in zip(datas, enc)])
    assert (st == 0).all() and outs == datas


@pytest.mark.gpu
def test_gpu_long_four_way_streams_take_the_chain_kernel(engine, norc):
    """4-way streams of >= 16 KiB go to rans4x16_big.hip (one stream per wavefront, scalar renormalisation window, one-read order-0 table, the LDS forms of the
    order-1 tables) -- lengths around the hand-over and every remainder mod 4, every table form: dense (qual4 / bases), bucket (qual41), lists that do not
    fit the pool (bytes: 256 contexts of 256 symbols), a single-symbol alphabet, order 0 and 1, raw (size in the stream), NOSZ, and behind PACK / RLE /
    STRIPE (pre-parsed descriptors); then damaged streams: the verdict is the oracle's."""
    rng = np.random.default_rng(404)
    blocks, want = [], []
    for kind in ("qual4", "qual41", "bases", "bytes", "const"):
        for n in (16383, 16384, 16385, 16386, 16387, 40_001, 262_146, 1_000_003):
            d = synth_series(rng, kind, n)
            for fl in (0, 1, 0x10, 0x11, 0x80, 0x81, 0x41, 0xC1, 0x09, 0x08):
                if n > 300_000 and fl not in (0, 1, 0x81): continue
                blocks.append((5, norc.encode(d, fl), len(d))); want.append(d)
    outs, st = engine.cram_uncompress_blocks(blocks)
    bad = [(i, hex(blocks[i][1][0]), blocks[i][2]) for i in range(len(blocks)) if st[i] != 0 or outs[i] != want[i]]
    assert not bad, bad[:10]
    small = synth_series(rng, "qual41", 30_000)
    base = [norc.encode(small, fl) for fl in (0, 1)] + [norc.encode(synth_series(rng, "qual4", 30_000), 1), norc.encode(synth_series(rng, "bytes", 30_000), 1)]
    dmg = []
    for rep in range(600):
        b = bytearray(base[rep & 3])
        pos = int(rng.integers(1, len(b)))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        dmg.append(bytes(b))
    dmg += [base[0][:-5], base[1][:-1], base[1][:len(base[1]) // 2], base[2][:200]]
    outs, st = engine.cram_uncompress_blocks([(5, b, 30_000) for b in dmg])
    for b, o, s in zip(dmg, outs, st):
        rc, w = norc.decode(b, 30_000)
        if rc == 0 and len(w) == 30_000: assert s == 0 and o == w
        else: assert s != 0
