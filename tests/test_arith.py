"""Adaptive arithmetic ("range") coder, CRAM 3.1 block method 6 -- PARITY UNPINNED (see
oracle/arith_oracle.c): htscodecs is absent from the reference and no method-6 stream exists in its
tests.  CPU part: the oracle's encoder/decoder agree over every flag set; GPU part: the gfx950 decoder is
bit-exact with the oracle, the gfx950 encoder is byte-identical to it."""
import numpy as np
import pytest

from tests import refutil
from tests.test_rans4x8 import synth_series
from tests.test_ransnx16 import runs_series

# the RANS_PR*-style sets htslib passes for arith ({1,64,9,128,129,192,193}, cram_io.c:1877) and more
ALL_FLAGS = [0, 1, 64, 65, 9, 8, 0x48, 128, 129, 192, 193, 0x20, 0xA0, 0x10, 0x11, 0x51, 0x90]
SIZES = (0, 1, 2, 3, 4, 5, 8, 63, 64, 65, 100, 1000, 4097, 150_000)


@pytest.fixture(scope="module")
def aorc(built):
    return refutil.ArithOracle()


@pytest.mark.parametrize("kind", ["qual4", "qual41", "bases", "bytes", "const", "runs"])
def test_oracle_roundtrip_all_flags(aorc, kind):
    rng = np.random.default_rng(len(kind) + 100)
    for n in SIZES:
        d = runs_series(rng, n) if kind == "runs" else synth_series(rng, kind, n)
        for fl in ALL_FLAGS:
            e = aorc.encode(d, fl)
            rc, out = aorc.decode(e, len(d), len(d) if fl & 0x10 else -1)
            assert rc == 0 and out == d, (kind, n, hex(fl))
            if n >= 1000 and not (e[0] & 0x20) and kind != "const":
                assert aorc.decode(e[:len(e) // 2], len(d), len(d))[0] == -1     # truncation is detected


def test_oracle_models_actually_compress(aorc):
    rng = np.random.default_rng(3)
    q = synth_series(rng, "qual4", 300_000)                     # 4-state Markov chain: order 1 pays
    o0, o1 = len(aorc.encode(q, 0)), len(aorc.encode(q, 1))
    assert o1 < 0.8 * o0 < 0.8 * 0.5 * len(q)
    r = runs_series(rng, 300_000, nsym=6, mean=40)
    assert len(aorc.encode(r, 64)) < 0.5 * len(aorc.encode(r, 0))
    assert len(aorc.encode(bytes(100_000), 65)) < 200
    b = synth_series(rng, "bases", 100_000)
    assert aorc.encode(b, 128)[0] == 128 and len(aorc.encode(b, 128)) < 0.3 * len(b)
    assert aorc.encode(q, 0)[:1] == b"\x00" and aorc.encode(q, 0)[4] == max(q) + 1    # flags, size(3), max_sym


@pytest.mark.gpu
def test_gpu_decoder_matches_oracle(engine, aorc):
    rng = np.random.default_rng(11)
    blocks, want = [], []
    for kind in ("qual4", "qual41", "bases", "bytes", "const", "runs"):
        for n in SIZES:
            d = runs_series(rng, n) if kind == "runs" else synth_series(rng, kind, n)
            for fl in ALL_FLAGS:
                blocks.append((6, aorc.encode(d, fl), len(d))); want.append(d)
    big = synth_series(rng, "qual41", 1_000_000)                     # crosses many model halvings
    for fl in (0, 1, 65, 9, 193):
        blocks.append((6, aorc.encode(big, fl), len(big))); want.append(big)
    wide = bytes(rng.integers(0, 200, 300_000, dtype=np.uint8))      # order-1 with 200 symbols: models in global scratch
    mid = bytes(rng.integers(0, 100, 300_000, dtype=np.uint8))       # order-1 with 100 symbols: big LDS pool
    for d in (wide, mid):
        for fl in (1, 65):
            blocks.append((6, aorc.encode(d, fl), len(d))); want.append(d)
    outs, st = engine.cram_uncompress_blocks(blocks)
    bad = [(i, hex(blocks[i][1][0]), blocks[i][2], int(st[i])) for i in range(len(blocks)) if st[i] != 0 or outs[i] != want[i]]
    assert not bad, bad[:10]
    # bzip2 payload (EXT) is reported as unsupported, not mis-decoded
    outs, st = engine.cram_uncompress_blocks([(6, bytes([0x04, 5]) + b"BZh91", 5)])
    assert list(st) == [-3]


@pytest.mark.gpu
def test_gpu_decoder_fuzz_agrees_with_oracle(engine, aorc):
    rng = np.random.default_rng(12)
    small = synth_series(rng, "qual41", 20_000)
    runs = (small[:200] + bytes([70]) * 300) * 40
    base = [aorc.encode(small, fl) for fl in (0, 1, 9, 128)] + [aorc.encode(runs, fl) for fl in (64, 65, 0x48, 193)]
    bad = []
    for rep in range(400):
        b = bytearray(base[rep & 7])
        pos = int(rng.integers(1, len(b)))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(b))
    bad += [base[0][:-5], base[1][:40], b"", b"\x01"]
    outs, st = engine.cram_uncompress_blocks([(6, b, len(small)) for b in bad])
    for b, o, s in zip(bad, outs, st):
        rc, want = aorc.decode(b, len(small), len(small))
        if rc == 0 and len(want) == len(small):
            assert s == 0 and o == want
        else:
            assert s != 0


@pytest.mark.gpu
def test_gpu_encoder_is_byte_identical_to_oracle(engine, aorc):
    rng = np.random.default_rng(13)
    datas, flags = [], []
    for kind in ("qual4", "qual41", "bases", "bytes", "const", "runs"):
        for n in SIZES:
            d = runs_series(rng, n) if kind == "runs" else synth_series(rng, kind, n)
            for fl in ALL_FLAGS:
                datas.append(d); flags.append(fl)
    wide = bytes(rng.integers(0, 200, 100_000, dtype=np.uint8))
    mid = bytes(rng.integers(0, 100, 100_000, dtype=np.uint8))
    for d in (wide, mid):
        for fl in (1, 65, 9):
            datas.append(d); flags.append(fl)
    enc = engine.arith_encode_host(datas, flags)
    bad = [(len(d), hex(fl)) for d, fl, e in zip(datas, flags, enc) if e != aorc.encode(d, fl)]
    assert not bad, bad[:12]
    outs, st = engine.cram_uncompress_blocks([(6, e, len(d)) for d, e in zip(datas, enc)])
    assert (st == 0).all() and outs == datas


@pytest.mark.gpu
def test_gpu_tiny_alphabets_starting_at_zero(engine, aorc):
    """Alphabets {0}, {0,1}, {0,1,2}: the first symbols are coded with totals of 1 .. 3, where range / total reaches 2^30 .. 2^32 - 1 -- the short division
    of the decoder (arith_dev.h udiv_small_quotient) mis-read such remainders until round 4 (an all-zero block did not decode)."""
    rng = np.random.default_rng(3)
    datas, flags = [], []
    for m in (1, 2, 3):
        for n in (1, 2, 5, 64, 1000, 70_000, 300_000):
            for rep in range(6 if n <= 1000 else 1):
                d = bytes(rng.integers(0, m, n, dtype=np.uint8)) if rep else bytes(n)
                for fl in (0, 1, 64, 65, 9, 193):
                    datas.append(d); flags.append(fl)
    enc = engine.arith_encode_host(datas, flags)
    bad = [(len(d), max(d) + 1, hex(fl)) for d, fl, e in zip(datas, flags, enc) if e != aorc.encode(d, fl)]
    assert not bad, bad[:12]
    outs, st = engine.cram_uncompress_blocks([(6, e, len(d)) for d, e in zip(datas, enc)])
    bad = [(len(d), max(d) + 1, hex(fl), int(s)) for d, fl, o, s in zip(datas, flags, outs, st) if s != 0 or o != d]
    assert not bad, bad[:12]


def two_phase_cases():
    """streams for the two-phase encoder (arith_enc2.hip): sizes around its 64-position steps, alphabets of 1 / 2 / 40 / 64 / 65 / 130 / 256 symbols (register models up
    to 64, the LDS form above), skew that halves the models many times, order 0 and 1 with STRIPE / PACK / RLE around it, and -- for the run lists of the RLE form --
    runs of every length around the part boundaries (3, 6, 9), runs that span many steps, a stream that is one run"""
    rng = np.random.default_rng(77)
    datas, flags = [], []
    for n in (8192, 8193, 8255, 8256, 8257, 20_000, 300_000):
        for m in (1, 2, 40, 64, 65, 130, 256):
            p = rng.dirichlet(np.full(m, 0.3)) if m > 1 else np.ones(1)
            d = bytes(rng.choice(m, n, p=p).astype(np.uint8))
            if m == 256: d = bytes([255]) + d[1:]
            for fl in (0, 1) if n != 300_000 else (0, 1, 9, 8, 64, 65, 128, 129, 193):
                datas.append(d); flags.append(fl)
    top = bytes(rng.choice(np.array([7, 9, 200], dtype=np.uint8), 200_000, p=[0.98, 0.015, 0.005]))     # one context carries nearly everything
    for fl in (0, 1, 65):
        datas.append(top); flags.append(fl)
    small = []
    for m in (1, 2, 3, 40):
        for n in (1, 2, 3, 5, 63, 64, 65, 127, 128, 129, 1000, 4097):
            small.append(bytes(rng.integers(0, m, n, dtype=np.uint8)))
    runs = [bytes(70_000), bytes([5]) * 64, bytes([5]) * 65 + bytes([6]),
            b"".join(bytes([int(s)]) * int(l) for s, l in zip(rng.integers(0, 4, 4000), rng.geometric(0.08, 4000))),           # lengths 1 .. ~100
            b"".join(bytes([i % 3]) * l for i, l in enumerate(list(range(1, 14)) * 40)),                                        # every length 1 .. 13
            b"".join(bytes([i % 7]) * int(l) for i, l in enumerate(rng.integers(60, 700, 300))),                                # runs across several steps
            b"".join(bytes([int(s)]) * int(l) for s, l in zip(rng.integers(0, 40, 30000), rng.geometric(0.5, 30000)))]          # many short runs, 40 symbols
    for d in small + runs:
        for fl in (0, 1, 64, 65, 193):
            datas.append(d); flags.append(fl)
    return datas, flags


@pytest.mark.gpu
def test_gpu_two_phase_encoder_is_byte_identical_to_oracle(engine, aorc, monkeypatch):
    """The two-phase encoder (events sorted by model, one task per model or bundle of models, one scalar coder pass per stream) against the one-pass kernels and the
    oracle.  By default it takes the order-1 / RLE streams from 4 KiB; HG_ARITH_2P_MIN=1 sends EVERY stream through it (order 0 and the tiny ones too), HG_ARITH_2P=0 none."""
    datas, flags = two_phase_cases()
    monkeypatch.setenv("HG_ARITH_2P_MIN", "1")                          # (read by the library at every call)
    enc = engine.arith_encode_host(datas, flags)
    monkeypatch.delenv("HG_ARITH_2P_MIN")
    bad = [(len(d), len(set(d)), hex(fl)) for d, fl, e in zip(datas, flags, enc) if e != aorc.encode(d, fl)]
    assert not bad, bad[:12]
    assert enc == engine.arith_encode_host(datas, flags)                # the default threshold: same bytes
    monkeypatch.setenv("HG_ARITH_2P", "0")
    assert enc == engine.arith_encode_host(datas, flags)                # one pass: same bytes
    monkeypatch.delenv("HG_ARITH_2P")
    outs, st = engine.cram_uncompress_blocks([(6, e, len(d)) for d, e in zip(datas, enc)])
    bad = [(len(d), len(set(d)), hex(fl), int(s)) for d, fl, o, s in zip(datas, flags, outs, st) if s != 0 or o != d]
    assert not bad, bad[:12]


@pytest.mark.gpu
def test_gpu_short_divisions_are_exact(engine):
    """arith_dev.h replaces the two 32-bit divisions of a coder step (range / total, code / r) by single-precision estimates with one correction; checked here
    against the exact division on the device over 1.5 G random operand pairs in the coder's ranges (+ the corners): any disagreement would change a stream"""
    import ctypes as C
    from htslib_amd import _native as nat
    nat.lib.hg_debug_udiv_check.restype = C.c_long
    nat.lib.hg_debug_udiv_check.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    assert nat.lib.hg_debug_udiv_check(engine._h, 12345, 2000) == 0
    assert nat.lib.hg_debug_udiv_check(engine._h, 987654321, 2000) == 0
