"""Read-name tokeniser "tok3", CRAM 3.1 block method 8 -- PARITY UNPINNED (see oracle/tok3_oracle.c):
htscodecs is absent from the reference and no method-8 stream exists in its tests.  CPU part: the
oracle's encoder/decoder agree on the kinds of names the reference's own test data carries; GPU part:
the gfx950 decoder is bit-exact with the oracle and the gfx950 encoder byte-identical to it."""
import numpy as np
import pytest

from tests import refutil


def illumina_names(rng, n, paired=False):
    """SIM:1:FC01:<lane>:<tile>:<x>:<y>, coordinate-ish order; paired -> each name twice."""
    lane = np.sort(rng.integers(1, 9, n))
    tile = rng.integers(1101, 2679, n)
    x = rng.integers(1000, 30000, n)
    y = np.cumsum(rng.integers(0, 40, n)) + 1000
    out = []
    for i in range(n):
        nm = b"SIM:1:FC01:%d:%d:%d:%d" % (lane[i], tile[i], x[i], y[i])
        out.append(nm)
        if paired:
            out.append(nm)
    return b"".join(nm + b"\0" for nm in out)


def odd_names(rng, n):
    """Leading zeros, long digit runs, punctuation, empty names, very many tokens."""
    pool = [b"", b"a", b"0", b"007", b"r0001/1", b"r0002/2", b"ERR000123.45 len=150", b"x" * 300, b"1234567890123456789",
            b"a.b.c.d.e.f.g.h.i.j.k.l.m.n.o.p.q.r.s.t.u.v.w.x.y.z." * 6, b"read_000099", b"read_000100", b"read_000355",
            b"@@@@", b"00", b"000", b"4294967295", b"4294967296", b"999999999", b"0999999999", b"\xff\xfe\x01"]
    out = [pool[int(k)] for k in rng.integers(0, len(pool), n)]
    return b"".join(nm + b"\0" for nm in out)


def sample_sets(rng):
    return [b"", b"\0", b"a\0", illumina_names(rng, 1), illumina_names(rng, 2), illumina_names(rng, 300),
            illumina_names(rng, 5000), illumina_names(rng, 2000, paired=True), odd_names(rng, 400),
            b"".join(b"q%07d\0" % i for i in range(3000)), b"same\0" * 1000]


@pytest.fixture(scope="module")
def torc(built):
    return refutil.Tok3Oracle()


@pytest.mark.parametrize("use_arith", [0, 1])
def test_oracle_roundtrip(torc, use_arith):
    rng = np.random.default_rng(70 + use_arith)
    for d in sample_sets(rng):
        e = torc.encode(d, use_arith)
        assert len(e) >= 9 and int.from_bytes(e[:4], "little") == len(d) and e[8] == use_arith
        assert int.from_bytes(e[4:8], "little") == d.count(b"\0")
        rc, out = torc.decode(e, len(d))
        assert rc == 0 and out == d
        if len(d) > 2000:
            assert torc.decode(e[:len(e) // 2], len(d))[0] == -1
            assert torc.decode(e, len(d) - 1)[0] == -1
    assert torc.encode(b"no terminator") == b""


def test_oracle_compresses_names(torc):
    rng = np.random.default_rng(9)
    d = illumina_names(rng, 10_000)
    for use_arith in (0, 1):
        e = torc.encode(d, use_arith)
        assert len(e) < 0.25 * len(d)                       # ~25 bytes a name -> a few bytes
    dup = illumina_names(rng, 5000, paired=True)
    assert len(torc.encode(dup)) < 0.15 * len(dup)              # the mate costs a DUP token
    # constant leading tokens cost one implied TYPE stream each, not a byte per name
    assert len(torc.encode(b"same\0" * 1000)) < 150


@pytest.mark.gpu
@pytest.mark.parametrize("use_arith", [0, 1])
def test_gpu_decoder_matches_oracle(engine, torc, use_arith):
    rng = np.random.default_rng(80 + use_arith)
    sets = sample_sets(rng) + [illumina_names(rng, 10_000), odd_names(rng, 3000)]
    blocks = [(8, torc.encode(d, use_arith), len(d)) for d in sets]
    outs, st = engine.cram_uncompress_blocks(blocks)
    bad = [(i, len(sets[i]), int(st[i])) for i in range(len(sets)) if len(sets[i]) and (st[i] != 0 or outs[i] != sets[i])]
    assert not bad, bad


@pytest.mark.gpu
def test_gpu_decoder_fuzz_agrees_with_oracle(engine, torc):
    rng = np.random.default_rng(90)
    plain = illumina_names(rng, 400) + odd_names(rng, 100)
    base = [torc.encode(plain, 0), torc.encode(plain, 1)]
    bad = []
    for rep in range(300):
        b = bytearray(base[rep & 1])
        pos = int(rng.integers(0, len(b)))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(b))
    bad += [base[0][:-7], base[0][:9], base[1][:200], b"", b"\x05\x00\x00\x00\x01\x00\x00\x00\x00"]
    outs, st = engine.cram_uncompress_blocks([(8, b, len(plain)) for b in bad])
    for b, o, s in zip(bad, outs, st):
        rc, want = torc.decode(b, len(plain))
        if rc == 0 and len(want) == len(plain):
            assert s == 0 and o == want
        else:
            assert s != 0


@pytest.mark.gpu
def test_gpu_encoder_is_byte_identical_to_oracle(engine, torc):
    rng = np.random.default_rng(95)
    sets = sample_sets(rng) + [illumina_names(rng, 10_000), odd_names(rng, 3000), b"not names"]
    datas = sets + sets
    ua = [0] * len(sets) + [1] * len(sets)
    enc = engine.tok3_encode_host(datas, ua)
    bad = [(i, len(d), u) for i, (d, u, e) in enumerate(zip(datas, ua, enc)) if e != torc.encode(d, u)]
    assert not bad, bad
    ok = [(8, e, len(d)) for d, e in zip(datas, enc) if e]
    outs, st = engine.cram_uncompress_blocks(ok)
    assert (st == 0).all() and [o for o in outs] == [d for d, e in zip(datas, enc) if e]
