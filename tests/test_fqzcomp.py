"""fqzcomp quality codec, CRAM 3.1 block method 7 -- PARITY UNPINNED (see oracle/fqzcomp_oracle.c): htscodecs is absent
from the reference and no method-7 stream exists in its tests (all CRAM fixtures are v3.0).  CPU part: the oracle's
encoder/decoder agree over every parameter-block feature (quality maps, position / delta / quality tables, fixed and
variable lengths, duplicates, reversed records, two parameter sets with and without a selector table) and its array
coder over the edge cases; GPU part: the gfx950 decoder is bit-exact with the oracle, alone and through the CRAM block
dispatcher (the boundary cram_uncompress_block uses, cram/cram_io.c:1684-1695)."""
import numpy as np
import pytest

from tests import refutil

F = refutil.FqzOracle


@pytest.fixture(scope="module")
def qorc(built):
    return refutil.FqzOracle()


def reads(rng, nrec, maxlen, fixed=True, nq=41, dup=0.0):
    """Quality strings the way a sequencer writes them: a slow random walk that decays along the read."""
    lens = np.full(nrec, maxlen, np.uint32) if fixed else rng.integers(1, maxlen + 1, nrec).astype(np.uint32)
    off = np.concatenate([[0], np.cumsum(lens.astype(np.int64))]).astype(np.int64)
    q = np.empty(int(off[-1]), np.uint8)
    for r in range(nrec):
        n = int(lens[r])
        walk = np.cumsum(rng.integers(-2, 3, n)) - np.arange(n) * (nq / (3.0 * max(n, 1)))
        q[off[r]:off[r + 1]] = np.clip(nq - 1 + walk, 0, nq - 1).astype(np.uint8)
        if r and dup and lens[r] == lens[r - 1] and rng.random() < dup:
            q[off[r]:off[r + 1]] = q[off[r - 1]:off[r]]
    return q.tobytes(), lens, (rng.integers(0, 2, nrec) * 16 + rng.integers(0, 2, nrec) * 128 + 1).astype(np.uint32)   # BAM flags


VARIANTS = [(s, o) for s in range(4) for o in (0, F.SEL, F.REV, F.DEDUP, F.SEL | F.REV | F.DEDUP, F.SEL | F.STAB | F.REV | F.DEDUP,
                                               F.NOQMAP, F.SEL | F.STAB | F.REV | F.DEDUP | F.NOQMAP)]


def corpus(rng, nrec=120, maxlen=100):
    for strat, opts in VARIANTS:
        for fixed in (True, False):
            for nq in (2, 4, 8, 41, 94):
                yield strat, opts, reads(rng, nrec, maxlen, fixed, nq, dup=0.3 if opts & F.DEDUP else 0.0)


def test_oracle_roundtrip_every_feature(qorc):
    rng = np.random.default_rng(5)
    seen_flags = set()
    for strat, opts, (q, lens, fl) in corpus(rng):
        e = qorc.encode(q, lens, fl, strat, opts)
        rc, out, ln = qorc.decode(e, len(q), len(lens))
        assert rc == 0 and out == q and (ln == lens).all(), (strat, opts)
        k = 1 + (e[0] >= 0x80) + (e[1] >= 0x80 if e[0] >= 0x80 else 0)      # uint7 size, then version and the global flags
        assert e[k] == 5
        seen_flags.add(e[k + 1])
        assert qorc.decode(e[:len(e) * 2 // 3], len(q))[0] == -1           # truncation is detected
        assert qorc.decode(e, len(q) - 1)[0] == -1                         # does not fit
    assert seen_flags == {0, 1, 4, 5, 7}                                  # MULTI_PARAM, HAVE_STAB, DO_REV in every combination the encoder makes


def test_oracle_models_actually_compress(qorc):
    rng = np.random.default_rng(6)
    q, lens, fl = reads(rng, 2000, 150, True, 41)
    sizes = [len(qorc.encode(q, lens, None, s, 0)) for s in range(4)]
    assert max(sizes) < 0.45 * len(q)                                      # an order-0 coder needs ~0.6 here
    q8, lens8, _ = reads(rng, 2000, 150, True, 8)
    assert qorc.encode(q8, lens8, None, 0, 0) != qorc.encode(q8, lens8, None, 0, F.NOQMAP)                # the quality map is used
    assert len(qorc.encode(q8, lens8, None, 0, 0)) < 0.3 * len(q8)
    d, dl, _ = reads(rng, 2000, 150, True, 41, dup=0.5)
    assert len(qorc.encode(d, dl, None, 0, F.DEDUP)) < 0.7 * len(qorc.encode(d, dl, None, 0, 0))


def test_array_coder_edges(qorc):
    rng = np.random.default_rng(7)
    cases = [np.zeros(256, np.uint32), np.arange(256, dtype=np.uint32), np.minimum(np.arange(1024) >> 3, 127).astype(np.uint32),
             np.repeat(np.arange(4, dtype=np.uint32), 256),                # four runs of 256: 255 + 1 each
             np.concatenate([np.zeros(1, np.uint32), np.ones(255, np.uint32)]),      # last run is exactly 255 (header note in the oracle)
             np.concatenate([np.zeros(514, np.uint32), np.full(510, 3, np.uint32)]),  # skipped values, last run 2 x 255
             np.sort(rng.integers(0, 40, 1024)).astype(np.uint32), np.sort(rng.integers(0, 255, 256)).astype(np.uint32)]
    for a in cases:
        b = qorc.store_array(a)
        used, back = qorc.read_array(b + b"\x07\x07\x07", len(a))          # trailing bytes belong to the next field
        assert used == len(b) and (back == a).all(), a[:8]
        assert len(b) < 300


@pytest.mark.gpu
def test_gpu_decoder_matches_oracle(engine, qorc):
    rng = np.random.default_rng(8)
    blocks, want = [], []
    for strat, opts, (q, lens, fl) in corpus(rng, nrec=60, maxlen=90):
        blocks.append((7, qorc.encode(q, lens, fl, strat, opts), len(q))); want.append(q)
    for n, ln in ((1, 1), (1, 63), (1, 64), (1, 65), (2, 64), (3, 128), (1, 5000), (40, 1000)):   # sizes around the 64-byte store
        for opts in (0, F.REV | F.DEDUP, F.SEL | F.REV | F.DEDUP):
            q, lens, fl = reads(rng, n, ln, False, 41, dup=0.3)
            blocks.append((7, qorc.encode(q, lens, fl, 0, opts), len(q))); want.append(q)
    big, blens, bfl = reads(rng, 3000, 150, True, 41, dup=0.05)           # crosses many model halvings
    for opts in (0, F.SEL | F.REV | F.DEDUP):
        blocks.append((7, qorc.encode(big, blens, bfl, 1, opts), len(big))); want.append(big)
    wide = bytes(rng.integers(0, 256, 20_000, dtype=np.uint8))            # 256 symbols: 4 x 64 model entries per read
    blocks.append((7, qorc.encode(wide, [20_000], None, 3, 0), len(wide))); want.append(wide)
    outs, st = engine.cram_uncompress_blocks(blocks)
    assert (st == 0).all(), np.nonzero(st)[0][:10]
    for k, (o, w) in enumerate(zip(outs, want)):
        assert o == w, k


@pytest.mark.gpu
def test_gpu_decoder_rejects_what_the_oracle_rejects(engine, qorc):
    rng = np.random.default_rng(9)
    q, lens, fl = reads(rng, 50, 100, False, 41)
    good = qorc.encode(q, lens, fl, 0, F.SEL | F.REV)
    bad = [good[:len(good) // 2], good[:20], b"", bytes([good[0], good[1], 4]) + good[3:], good[:2] + bytes([good[2] ^ 0xFF]) + good[3:]]
    blocks = [(7, good, len(q))] + [(7, b, len(q)) for b in bad] + [(7, good, len(q) + 1), (7, good, len(q) - 1)]
    outs, st = engine.cram_uncompress_blocks(blocks)
    assert st[0] == 0 and outs[0] == q
    for k in range(1, len(blocks)):
        assert st[k] == -1 and outs[k] is None, k
        assert blocks[k][2] != len(q) or qorc.decode(blocks[k][1], blocks[k][2])[0] != 0, k      # the oracle rejects the same streams
    # damaged streams FIRST in a launch, valid ones behind them (a wavefront takes several streams in turn: an early exit from the fast path must leave
    # nothing in flight that could land in the next stream's registers -- the decoder issues its model prefetches by hand)
    flipped = []
    for k in range(24):
        b = bytearray(good); pos = int(rng.integers(8, len(good))); b[pos] ^= 1 << int(rng.integers(0, 8)); flipped.append(bytes(b))
    many = [(7, b, len(q)) for b in flipped] * 3 + [(7, good, len(q))] * 200
    outs, st = engine.cram_uncompress_blocks(many)
    for k in range(len(flipped) * 3):
        rc, dec = qorc.decode(many[k][1], many[k][2])[:2]
        assert (st[k] == 0) == (rc == 0), k
        if st[k] == 0: assert outs[k] == dec, k
    for k in range(len(flipped) * 3, len(many)):
        assert st[k] == 0 and outs[k] == q, k
    # mixed with the other methods of a CRAM 3.1 slice: the dispatcher runs the families side by side
    mixed = [(7, good, len(q)), (0, b"abc", 3), (7, good, len(q))]
    outs, st = engine.cram_uncompress_blocks(mixed)
    assert (st == 0).all() and outs[0] == q and outs[1] == b"abc" and outs[2] == q


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["two-phase", "one-pass"])
def test_gpu_encoder_matches_oracle_and_decodes(engine, qorc, monkeypatch, form):
    """The gfx950 encoder emits exactly the oracle's bytes for the parameter choice both make (selector for preset 1, reversal when
    flags are given, duplicates when one record in ten repeats), and the gfx950 decoder takes them back -- in its default two-phase form (events sorted by
    model, register models, scalar coder pass) and in the one-pass form (HG_FQZ_2P=0), which remains for blocks the work memory cannot take."""
    if form == "one-pass": monkeypatch.setenv("HG_FQZ_2P", "0")
    rng = np.random.default_rng(10)
    datas, lens, flags, strats, want = [], [], [], [], []
    for strat in range(4):
        for fixed in (True, False):
            for nq in (2, 5, 8, 41, 94):
                for with_flags in (True, False):
                    for dup in (0.0, 0.3):
                        q, ln, fl = reads(rng, 50, 80, fixed, nq, dup)
                        datas.append(q); lens.append(ln); flags.append(fl if with_flags else None); strats.append(strat)
                        opts = F.DEDUP | (F.REV if with_flags else 0) | (F.SEL if with_flags and strat == 1 else 0)
                        want.append(qorc.encode(q, ln, fl if with_flags else None, strat, opts))
    for n, ln_ in ((1, 1), (1, 64), (2, 65), (1, 3000), (700, 151), (5000, 150), (3, 70000)):
        q, ln, fl = reads(rng, n, ln_, False, 41, 0.2)
        datas.append(q); lens.append(ln); flags.append(fl); strats.append(n % 4)
        want.append(qorc.encode(q, ln, fl, n % 4, F.DEDUP | F.REV | (F.SEL if n % 4 == 1 else 0)))
    got = engine.fqz_encode_host(datas, lens, flags, strats)
    for k, (g, w) in enumerate(zip(got, want)):
        assert g == w, (k, len(g), len(w))
    outs, st = engine.cram_uncompress_blocks([(7, g, len(d)) for g, d in zip(got, datas)])
    assert (st == 0).all() and all(o == d for o, d in zip(outs, datas))
    # refused inputs: record lengths that do not add up, an empty record, no slice information
    bad = engine.fqz_encode_host([datas[0], datas[0]], [lens[0][:-1], np.concatenate([lens[0][:-1], [0], lens[0][-1:]])], [None, None], [0, 0])
    assert bad == [b"", b""]


@pytest.mark.gpu
def test_auto_tuner_offers_fqzcomp_when_the_slice_is_known(engine, qorc):
    """cram_compress_block2(fd, s, b, ...) with use_fqz: the FQZ* bits of a quality block's method set are tried next to rANS / gzip when
    the slice's record lengths travel with the block (cram_io.c:1801-1825), learnt by the metrics object, and written as method 7."""
    import ctypes as C
    from htslib_amd import _native as nat
    rng = np.random.default_rng(12)
    M = lambda *ids: sum(1 << i for i in ids)
    qset = M(1, 5, 17, 7, 13, 14, 15)                   # GZIP, RANS_PR0, RANS_PR1, FQZ, FQZ_b, FQZ_c, FQZ_d  (cram_encode.c:864-869 at level > 6)
    m1, m2 = nat.lib.hg_cram_metrics_new(), nat.lib.hg_cram_metrics_new()
    A = C.cast(m1, C.POINTER(nat.CramMetrics)).contents
    Bm = C.cast(m2, C.POINTER(nat.CramMetrics)).contents
    profile = rng.integers(5, 38, 100)                   # qualities that depend on the cycle: a position context pays, an order-1 model does not
    for call in range(6):
        ln = np.full(400, 100, np.uint32); fl = (rng.integers(0, 2, 400) * 128 + 1).astype(np.uint32)
        q = (np.tile(profile, 400) + rng.integers(-1, 2, 40_000)).astype(np.uint8).tobytes()
        outs, used = engine.cram_compress_blocks_metrics([q, q], [m1, m2], [qset, qset], level=7, fqz=[(ln, fl), None])
        assert used[0] == 7 and used[1] in (1, 5), (call, used, [len(o) for o in outs])   # with the slice fqzcomp wins; without it the bits are dropped
        assert len(outs[0]) < 0.9 * len(outs[1])
        rc, back, lens = qorc.decode(outs[0], len(q), len(ln))            # the oracle reads what the tuner kept
        assert rc == 0 and back == q and (lens == ln).all()
        back, st = engine.cram_uncompress_blocks([(int(u), o, len(q)) for o, u in zip(outs, used)])
        assert (st == 0).all() and back == [q, q]
    assert A.method in (7, 13, 14, 15) and Bm.method in (1, 5, 17)        # learnt: an fqzcomp preset / gzip or an Nx16 order
    assert not (Bm.revised_method & M(7, 13, 14, 15)) and (A.revised_method & M(7, 13, 14, 15))
    nat.lib.hg_cram_metrics_free(m1); nat.lib.hg_cram_metrics_free(m2)


def test_oracle_roundtrip_with_more_than_two_parameter_sets(qorc):
    """the format allows 256 parameter sets (htscodecs writes one or two): the oracle's encoder can be asked for 2 + k sets chosen per record"""
    rng = np.random.default_rng(21)
    for extra in (1, 2, 3, 6):
        for fixed in (True, False):
            q, lens, fl = reads(rng, 200, 90, fixed, 41, dup=0.1)
            e = qorc.encode(q, lens, fl, 1, F.SEL | F.REV | F.DEDUP | (extra << 5))
            assert e[1 + (1 if len(q) < 128 else 2 if len(q) < 16384 else 3)] == 5                 # version byte behind the uint7 size
            rc, out, ln = qorc.decode(e, len(q), 200)
            assert rc == 0 and out == q and (ln == lens).all(), extra


@pytest.mark.gpu
def test_gpu_decoder_takes_any_number_of_parameter_sets(engine, qorc):
    """streams with more than two parameter sets were refused (and with no CPU codec the block was undecodable).  The wavefront keeps two sets in
    LDS and swaps the others in from a global overflow image as records select them: 3, 4, 5 and 8 sets, with and without a selector table, fixed and
    variable lengths, next to ordinary one- and two-set streams in the same batch"""
    rng = np.random.default_rng(22)
    blocks, want = [], []
    for extra in (1, 2, 3, 6):
        for opts in (F.SEL, F.SEL | F.REV | F.DEDUP, F.SEL | F.STAB | F.REV, F.SEL | F.NOQMAP):
            for fixed in (True, False):
                for nq in (4, 41):
                    q, lens, fl = reads(rng, 300, 80, fixed, nq, dup=0.2 if opts & F.DEDUP else 0.0)
                    blocks.append((7, qorc.encode(q, lens, fl, extra & 3, opts | (extra << 5)), len(q))); want.append(q)
    q, lens, fl = reads(rng, 100, 100, True, 41)
    blocks.append((7, qorc.encode(q, lens, fl, 0, 0), len(q))); want.append(q)
    blocks.append((7, qorc.encode(q, lens, fl, 1, F.SEL), len(q))); want.append(q)
    outs, st = engine.cram_uncompress_blocks(blocks)
    assert (st == 0).all(), np.nonzero(st)[0][:10]
    assert outs == want
