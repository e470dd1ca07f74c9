"""rANS 4x8 (CRAM 3.0 block method 4).  CPU part pins the oracle against the reference's CRAM
fixtures; GPU part (-m gpu) checks the gfx950 decoder, through the C ABI, against the same vectors
and against the oracle on seeded synthetic CRAM data series."""
import numpy as np
import pytest

from tests import refutil


@pytest.fixture(scope="module")
def rorc(built):
    return refutil.Rans4x8Oracle()


def synth_series(rng, kind, n):
    if kind == "qual4":      # NovaSeq-like 4-bin qualities as a Markov chain
        vals = np.array([2, 12, 23, 37], dtype=np.uint8)
        change = rng.random(n) < 0.1
        idx = np.maximum.accumulate(np.where(change, np.arange(n), 0))
        return vals[rng.integers(0, 4, n)][idx].tobytes()
    if kind == "qual41":
        return np.clip(rng.normal(34, 5, n), 2, 41).astype(np.uint8).tobytes()
    if kind == "bases":
        return np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, n, p=[.2495, .2495, .2495, .2495, .002])].tobytes()
    if kind == "bytes":
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == "const":
        return bytes([7]) * n
    raise ValueError(kind)


def test_oracle_decodes_reference_cram_fixture_blocks(rorc):
    cases = list(refutil.rans_golden_cases())
    assert len(cases) >= 40 and {c[4] for c in cases} == {0, 1}
    pinned = 0
    for name, comp, usize, expect, order in cases:
        rc, out = rorc.decode(comp)
        assert rc == 0 and len(out) == usize, name
        if expect is not None:
            assert out == expect, name           # plaintext derived from the .sam twin, no rANS involved
            pinned += 1
        for o in (0, 1):                          # our encoder's streams are decodable and lossless
            rc2, back = rorc.decode(rorc.encode(out, o))
            assert rc2 == 0 and back == out
    assert pinned >= 28                  # QS, RN, SC, BF, RL, AP, TS, one-byte and string aux tags derived from the .sam / .bam twins


@pytest.mark.parametrize("kind", ["qual4", "qual41", "bases", "bytes", "const"])
def test_oracle_roundtrip_and_malformed(rorc, kind):
    rng = np.random.default_rng(hash(kind) % 1000)
    for n in (0, 1, 3, 4, 5, 17, 1000, 150_000):
        d = synth_series(rng, kind, n)
        for order in (0, 1):
            e = rorc.encode(d, order)
            rc, out = rorc.decode(e)
            assert rc == 0 and out == d
            if n >= 1000:
                assert rorc.decode(e[:-7])[0] == -1                         # truncated
                assert rorc.decode(e[:1] + bytes([e[1] ^ 1]) + e[2:])[0] == -1   # size field lies


@pytest.mark.gpu
def test_gpu_decodes_reference_cram_fixture_blocks(engine, rorc):
    cases = list(refutil.rans_golden_cases())
    outs, st = engine.rans4x8_decode_host([c[1] for c in cases])
    assert (st == 0).all()
    for (name, comp, usize, expect, order), got in zip(cases, outs):
        assert len(got) == usize, name
        assert got == rorc.decode(comp)[1], name
        if expect is not None:
            assert got == expect, name


@pytest.mark.gpu
def test_gpu_matches_oracle_on_synthetic_series(engine, rorc):
    rng = np.random.default_rng(77)
    plains, streams = [], []
    for kind in ("qual4", "qual41", "bases", "bytes", "const"):
        for n in (0, 1, 2, 3, 4, 5, 6, 7, 8, 63, 64, 65, 1000, 4097, 150_000):
            d = synth_series(rng, kind, n)
            for order in (0, 1):
                plains.append(d); streams.append(rorc.encode(d, order))
    outs, st = engine.rans4x8_decode_host(streams)
    assert (st == 0).all()
    assert outs == plains


@pytest.mark.gpu
def test_gpu_rejects_malformed_streams_like_oracle(engine, rorc):
    rng = np.random.default_rng(5)
    d = synth_series(rng, "qual41", 20000)
    good0, good1 = rorc.encode(d, 0), rorc.encode(d, 1)
    bad = [good0, good0[:-9], good1[:len(good1) // 2], good1, b"", b"\x00" * 5,
           good0[:1] + bytes([good0[1] ^ 1]) + good0[2:], bytes([2]) + good0[1:]]
    want = [rorc.decode(b)[0] for b in bad]
    assert want == [0, -1, -1, 0, -1, -1, -1, -1]
    outs, st = engine.rans4x8_decode_host(bad)
    assert list(st) == want
    assert outs[0] == d and outs[3] == d
    # corrupt payload bytes: must never hang or fault; result = oracle's verdict, and equal bytes when it decodes
    for rep in range(200):
        b = bytearray(good1 if rep & 1 else good0)
        pos = int(rng.integers(9, len(b)))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        bad.append(bytes(b))
    outs, st = engine.rans4x8_decode_host(bad)
    for b, o, s in zip(bad, outs, st):
        rc, want_out = rorc.decode(b)
        assert s == rc
        if rc == 0:
            assert o == want_out


@pytest.mark.gpu
def test_gpu_encoder_is_byte_identical_to_oracle_and_decodes(engine, rorc):
    """The gfx950 rANS 4x8 encoder emits exactly the oracle's bytes (the oracle's DECODER being pinned
    on the reference's CRAM fixtures); re-encoded fixture blocks decode back on the GPU."""
    rng = np.random.default_rng(31)
    datas, orders = [], []
    for kind in ("qual4", "qual41", "bases", "bytes", "const"):
        for n in (0, 1, 2, 3, 4, 5, 6, 7, 8, 63, 64, 65, 1000, 4097, 150_000):
            d = synth_series(rng, kind, n)
            for o in (0, 1):
                datas.append(d); orders.append(o)
    for name, comp, usize, expect, order in refutil.rans_golden_cases():      # the fixtures' own plaintexts
        d = rorc.decode(comp)[1]
        datas += [d, d]; orders += [0, 1]
    enc = engine.rans4x8_encode_host(datas, orders)
    for d, o, e in zip(datas, orders, enc):
        assert e == rorc.encode(d, o), (len(d), o)
    outs, st = engine.rans4x8_decode_host(enc)
    assert (st == 0).all() and outs == datas
