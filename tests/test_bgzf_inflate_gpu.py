"""GPU parity tests (pytest -m gpu): the gfx950 BGZF inflate kernel, called through the C ABI,
against the oracle / the reference's golden vectors.  Bit-exact or it fails."""
import hashlib
import struct
import zlib

import numpy as np
import pytest

from tests import refutil
from htslib_amd import synth

pytestmark = pytest.mark.gpu


def fixed_block_with_far_match():
    """BFINAL=1 BTYPE=01, literal 'a', then <length 3, distance 5>: reaches before the block start."""
    bits = []
    def put(v, n, msb_first=False):
        for i in (range(n - 1, -1, -1) if msb_first else range(n)):
            bits.append((v >> i) & 1)
    put(1, 1); put(1, 2)
    put(0x30 + ord("a"), 8, True)          # literal 0..143: 8-bit codes 00110000 + value
    put(1, 7, True)                        # symbol 257 = length 3
    put(4, 5, True); put(0, 1)             # distance code 4 (+1 extra bit) = 5
    put(0, 7, True)                        # end of block
    while len(bits) % 8:
        bits.append(0)
    return bytes(sum(b << k for k, b in enumerate(bits[i:i + 8])) for i in range(0, len(bits), 8))


def gpu_inflate(engine, comp):
    from htslib_amd.bgzf import DeviceStream
    ds = DeviceStream(comp)
    ds.inflate(engine)
    return ds.result()


@pytest.mark.parametrize("name,comp,plain", list(refutil.golden_cases()), ids=lambda v: v if isinstance(v, str) else None)
def test_reference_fixtures_bit_exact(engine, name, comp, plain):
    got, st = gpu_inflate(engine, comp)
    assert (st == 0).all()
    assert got == plain
    # and through the synchronous host-buffer entry point
    got2, st2 = engine.bgzf_inflate_host(comp)
    assert got2 == plain and (st2 == 0).all()


@pytest.mark.parametrize("level", [0, 1, 2, 4, 6, 9])
def test_synthetic_bam_zlib_levels_vs_oracle(engine, oracle, level):
    plain, bg = synth.bam_bgzf(2 << 20, seed=synth.SEED + 11 * level, level=level)
    n, want = oracle.decompress(bg)
    assert n == len(plain) and want == plain
    got, st = gpu_inflate(engine, bg)
    assert (st == 0).all() and got == want


@pytest.mark.skipif(not refutil.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("level", [1, 6, 9])
def test_libdeflate_streams_from_real_reference(engine, oracle, level):
    """libdeflate splits a BGZF payload into several deflate blocks with fresh tables."""
    plain, _ = synth.bam_bgzf(3 << 20, seed=synth.SEED + 100 + level)
    comp = refutil.ref_bgzip(["-l", str(level)], plain, "libdeflate")
    got, st = gpu_inflate(engine, comp)
    assert (st == 0).all() and got == plain


def test_fastq_and_strategies(engine, oracle):
    fq = synth.fastq(400_000)
    rng = np.random.default_rng(3)
    datas = {"fastq": fq[:65280], "zeros": bytes(65280), "rand": rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(),
             "ab": b"ab" * 30000, "one": b"x", "max": bytes(rng.integers(0, 4, 65536, dtype=np.uint8))}
    blocks, want = [], []
    for d in datas.values():
        for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            for lvl in (1, 9):
                b = refutil.raw_block(d, lvl, strat)
                if len(b) <= 65536:
                    blocks.append(b); want.append(d)
    comp = b"".join(blocks)
    n, ow = oracle.decompress(comp)
    assert ow == b"".join(want)
    got, st = gpu_inflate(engine, comp)
    assert (st == 0).all() and got == ow


def test_empty_ragged_and_eof_blocks(engine, oracle):
    # empty stream
    got, st = gpu_inflate(engine, b"")
    assert got == b"" and len(st) == 0
    # EOF marker only, EOF markers embedded mid-stream (append mode, test/test_bgzf.c:511-582)
    parts, want = [], []
    rng = np.random.default_rng(5)
    for i in range(300):
        n = int(rng.integers(0, 700)) if i % 7 else 0
        d = rng.integers(65, 70, n, dtype=np.uint8).tobytes()
        parts.append(synth.bgzf_block(d, level=int(rng.integers(0, 10)))); want.append(d)
    comp = b"".join(parts) + synth.BGZF_EOF
    got, st = gpu_inflate(engine, comp)
    assert (st == 0).all() and got == b"".join(want)
    assert oracle.decompress(comp)[1] == got
    got, st = gpu_inflate(engine, synth.BGZF_EOF)
    assert got == b"" and list(st) == [0]


def test_error_codes_match_bgzf_uncompress(engine, oracle):
    """Per-block status follows bgzf_uncompress (bgzf.c:730-804): -1 inflate failure, -2 CRC."""
    data = synth.fastq(120_000)[:60000]
    good = refutil.raw_block(data)
    payload = good[18:-8]
    cases = {
        "good": good,
        "bad_crc": good[:-8] + bytes([good[-8] ^ 0x40]) + good[-7:],
        "truncated": refutil.wrap_payload(payload[:len(payload) // 2], data),
        "btype3": refutil.wrap_payload(b"\x07" + payload[1:], data),
        "stored_bad_nlen": refutil.wrap_payload(b"\x01\x05\x00\x00\x00hello", b"hello"),
        "dist_too_far": refutil.wrap_payload(fixed_block_with_far_match(), b"aaaa"),
        "good_again": refutil.raw_block(b"tail " * 1000),
    }
    comp = b"".join(cases.values())
    want = [oracle.uncompress_block(b)[0] for b in cases.values()]
    assert want == [0, -2, -1, -1, -1, -1, 0]
    from htslib_amd import _native as nat
    from htslib_amd.bgzf import DeviceStream
    ds = DeviceStream(comp)
    ds.inflate(engine)
    got, st = ds.result()
    assert list(st) == want
    # good blocks are still delivered intact around the failures
    assert got[:len(data)] == data and got.endswith(b"tail " * 1000)
    with pytest.raises(nat.HgError) as e:
        engine.bgzf_inflate_host(comp)
    assert e.value.code == -6 and engine.last_bad == (1, -2)


def test_bitflip_fuzz_status_parity_with_oracle(engine, oracle):
    """Corrupt one bit in each of 400 blocks: the kernel must never hang or write out of
    bounds, and must classify every block exactly like the oracle (0 / -1 / -2)."""
    plain, bg = synth.bam_bgzf(1 << 20, seed=99)
    blocks = refutil.split_blocks(bg)
    rng = np.random.default_rng(12345)
    out, want_rc, want_data = [], [], []
    for rep in range(400):
        off, clen, isize = blocks[rep % len(blocks)]
        b = bytearray(bg[off:off + clen])
        pos = int(rng.integers(18, clen - 4))          # payload or CRC, keep framing + ISIZE intact
        b[pos] ^= 1 << int(rng.integers(0, 8))
        b = bytes(b)
        rc, d = oracle.uncompress_block(b)
        out.append(b); want_rc.append(rc); want_data.append(d if rc == 0 else None)
    comp = b"".join(out)
    got, st = gpu_inflate(engine, comp)
    assert list(st) == want_rc
    assert -1 in want_rc and -2 in want_rc
    pos = 0
    for (b, rc, d) in zip(out, want_rc, want_data):
        isize = struct.unpack("<I", b[-4:])[0]
        if rc == 0:
            assert got[pos:pos + isize] == d
        pos += isize


def test_large_stream_checksum_property(engine):
    """256 MiB: every block CRC-checked in-kernel + md5 of the whole plain image."""
    import bench
    comp = bench.prepare(synth.SEED, 256 << 20, 6, 32, None)
    got, st = gpu_inflate(engine, comp)
    assert (st == 0).all()
    # independent decoder for the expected image
    import gzip
    want = gzip.decompress(comp)
    assert hashlib.md5(got).hexdigest() == hashlib.md5(want).hexdigest()


def test_crc32_batch(engine, oracle):
    from htslib_amd.bgzf import crc32_device
    rng = np.random.default_rng(1)
    sizes = [0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1023, 4096, 65279, 65280, 65536, 1 << 20, (1 << 20) + 3]
    bufs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in sizes]
    got = crc32_device(engine, bufs)
    assert [int(x) for x in got] == [zlib.crc32(b) for b in bufs] == [oracle.crc32(b) for b in bufs]


def test_declared_size_lies_and_chunk_boundaries(engine, oracle):
    """The decoder writes through a 1 KiB ring and flushes 256-byte chunks: output that stops short of / runs past the declared ISIZE
    must give the reference's verdict (bgzf.c:775-804) and never touch a neighbour's bytes; lengths around every chunk boundary decode."""
    rng = np.random.default_rng(11)
    text = synth.fastq(200_000)
    blocks, want = [], []
    for n in [1, 2, 3, 4, 63, 64, 65, 255, 256, 257, 258, 259, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, 1279, 1280, 4095, 4096, 4097, 65279, 65280]:
        for kind in range(3):
            d = text[:n] if kind == 0 else bytes(n) if kind == 1 else rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            blocks.append(refutil.raw_block(d, 6)); want.append(d)
    comp = b"".join(blocks)
    got, st = gpu_inflate(engine, comp)
    assert (st == 0).all() and got == b"".join(want) == oracle.decompress(comp)[1]
    # ISIZE says less / more than the stream makes; a good neighbour on either side must survive
    data = text[:50_000]
    payload = refutil.raw_block(data, 6)[18:-8]
    zeros = refutil.raw_block(bytes(60_000), 6)[18:-8]                       # 258-byte overlapping copies, two chunks per copy
    guard = refutil.raw_block(b"guard " * 2000)
    cases = [guard,
             refutil.wrap_payload(payload, data, isize=len(data) - 1), guard,
             refutil.wrap_payload(payload, data, isize=len(data) - 300), guard,
             refutil.wrap_payload(payload, data, isize=1000), guard,
             refutil.wrap_payload(payload, data, isize=len(data) + 1), guard,
             refutil.wrap_payload(payload, data, isize=65536), guard,
             refutil.wrap_payload(zeros, bytes(60_000), isize=59_000), guard,
             refutil.wrap_payload(zeros, bytes(60_000), isize=100), guard,
             refutil.wrap_payload(zeros, bytes(60_000)), guard]
    comp = b"".join(cases)
    want_st = [oracle.uncompress_block(b)[0] for b in cases]
    assert want_st[0::2] == [0] * len(cases[0::2]) and want_st[-2] == 0 and all(w != 0 for w in want_st[1:-2:2])
    from htslib_amd.bgzf import DeviceStream
    ds = DeviceStream(comp)
    ds.inflate(engine)
    got, st = ds.result()
    assert list(st) == want_st
    # every good block is intact wherever it sits (offsets follow the declared sizes)
    off = 0
    for b, w in zip(cases, want_st):
        isize = struct.unpack("<I", b[-4:])[0]
        if w == 0:
            exp = b"guard " * 2000 if b is guard else bytes(60_000)
            assert got[off:off + isize] == exp
        off += isize
