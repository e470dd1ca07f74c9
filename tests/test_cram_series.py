"""SURVEY §8f N2, first step: CRAM integer data series <-> EXTERNAL blocks (ITF8), whole blocks at a time.

Reference: cram_external_decode_int (cram/cram_codecs.c:350-368) -> safe_itf8_get (cram/cram_io.c:644-673);
cram_external_encode_int (cram_codecs.c:523-527) -> itf8_put (cram_io.c:277-305).

CPU: the oracle (oracle/cram_series_oracle.c) equals the reference's OWN two functions, spliced from cram_io.c into a scratch
program at test time (tests/native/gen_itf8_ref.sh), on random and edge-case blocks -- values, counts, consumed bytes, error
flag, encoded bytes.  GPU: hg_cram_itf8_decode_host / _encode_host equal the oracle on the same blocks, and the BF / RL / AP
blocks of the reference's CRAM fixtures (plaintext derived from the .sam twins, tests/golden/rans4x8) decode to the twins' values."""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import refutil

ROOT = refutil.ROOT
REF = "/root/reference"


def orc():
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    L.orc_itf8_decode_block.restype = C.c_long
    L.orc_itf8_decode_block.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orc_itf8_encode_block.restype = C.c_size_t
    L.orc_itf8_encode_block.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
    L.orc_byte_array_stop_split.restype = C.c_long
    L.orc_byte_array_stop_split.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, C.c_void_p, C.c_size_t]
    return L


def orc_decode(L, b: bytes, cap=None):
    cap = len(b) + 1 if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=np.int32)
    nok = C.c_size_t(0)
    rc = L.orc_itf8_decode_block(b, len(b), out.ctypes.data, cap, C.byref(nok))
    return rc, out[:nok.value].copy()


def orc_encode(L, vals: np.ndarray) -> bytes:
    vals = np.ascontiguousarray(vals, dtype=np.int32)
    out = C.create_string_buffer(5 * len(vals) + 8)
    n = L.orc_itf8_encode_block(vals.ctypes.data, len(vals), out)
    return out.raw[:n]


def value_sets():
    rng = np.random.default_rng(424242)
    sets = [np.array([], dtype=np.int32), np.array([0], dtype=np.int32),
            np.array([0, 127, 128, 16383, 16384, 2097151, 2097152, 268435455, 268435456, 2**31 - 1, -1, -2**31, -129], dtype=np.int32)]
    for n in (1, 63, 64, 65, 1000, 4095, 4096, 4097, 50000):
        sets.append(rng.integers(0, 128, n).astype(np.int32))                                     # one byte each: flags, read lengths
    for n in (819, 820, 10007):
        sets.append(np.full(n, -1, dtype=np.int32))                                               # five bytes each: tiles start mid-value
    for n in (5000, 200000):
        k = rng.integers(0, 5, n)
        hi = np.array([7, 14, 21, 28, 32])[k]
        v = (rng.integers(0, 2**62, n, dtype=np.int64) & ((1 << hi.astype(np.int64)) - 1)).astype(np.uint32)
        sets.append(v.view(np.int32))                                                             # every length, randomly mixed
    d = np.cumsum(rng.integers(0, 300, 30000)).astype(np.int32)
    sets.append(np.diff(d, prepend=0).astype(np.int32))                                           # AP deltas
    return sets


def test_oracle_equals_reference_functions(built, tmp_path):
    if not os.path.isdir(REF):
        pytest.skip("no reference checkout here")
    subprocess.run(["bash", os.path.join(ROOT, "tests", "native", "gen_itf8_ref.sh"), str(tmp_path)], check=True)
    exe = str(tmp_path / "itf8_ref")
    L = orc()
    for vals in value_sets():
        if len(vals) > 60000:
            vals = vals[:60000]
        enc = subprocess.run([exe, "e"], input=vals.tobytes(), capture_output=True, check=True).stdout
        assert enc == orc_encode(L, vals)
        for cut in (0, 1, 2, 3, 4):                                                               # whole block, then ends inside the last value
            b = enc[:len(enc) - cut] if cut else enc
            r = subprocess.run([exe, "d"], input=b, capture_output=True, check=True).stdout.decode().split("\n")
            end = [x for x in r if x.startswith("END")][0].split()
            ref_vals = np.array([int(x) for x in r if x and not x.startswith("END")], dtype=np.int64).astype(np.int32)
            rc, got = orc_decode(L, b)
            assert (rc < 0) == (int(end[2]) != 0), (len(vals), cut)
            assert np.array_equal(got, ref_vals), (len(vals), cut)
            if rc >= 0:
                assert int(end[1]) == len(b)
    # arbitrary bytes are a valid block up to a possibly cut last value
    rng = np.random.default_rng(99)
    for n in (1, 7, 300, 5000):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        r = subprocess.run([exe, "d"], input=b, capture_output=True, check=True).stdout.decode().split("\n")
        ref_vals = np.array([int(x) for x in r if x and not x.startswith("END")], dtype=np.int64).astype(np.int32)
        rc, got = orc_decode(L, b)
        assert np.array_equal(got, ref_vals) and (rc < 0) == (r[-2].split()[2] != "0")


def fixture_columns():
    """(name, plaintext of the block, expected values) for the BF / RL / AP blocks whose plaintext comes from the .sam twins"""
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "rans4x8", "MANIFEST.json")))
    out = []
    for name, e in sorted(man.items()):
        if e["series"] in ("BF", "RL", "AP") and e["expected_hex"]:
            b = bytes.fromhex(e["expected_hex"])
            vals, p = [], 0
            while p < len(b):                                                                     # python ITF8 reader (as in make_golden_rans.py)
                v = b[p]
                if v < 0x80: x, p = v, p + 1
                elif v < 0xC0: x, p = ((v & 0x3F) << 8) | b[p + 1], p + 2
                elif v < 0xE0: x, p = ((v & 0x1F) << 16) | (b[p + 1] << 8) | b[p + 2], p + 3
                elif v < 0xF0: x, p = ((v & 0x0F) << 24) | (b[p + 1] << 16) | (b[p + 2] << 8) | b[p + 3], p + 4
                else: x, p = ((v & 0x0F) << 28) | (b[p + 1] << 20) | (b[p + 2] << 12) | (b[p + 3] << 4) | (b[p + 4] & 0x0F), p + 5
                vals.append(x - (1 << 32) if x >= 1 << 31 else x)
            out.append((name, b, np.array(vals, dtype=np.int32)))
    return out


def test_oracle_on_fixture_columns(built):
    L = orc()
    cols = fixture_columns()
    assert len(cols) >= 9
    for name, b, vals in cols:
        rc, got = orc_decode(L, b)
        assert rc == len(vals) and np.array_equal(got, vals), name
        assert orc_encode(L, vals) == b, name


def gpu_decode(nat, eng, blocks, caps=None):
    n = len(blocks)
    caps = [len(b) + 1 for b in blocks] if caps is None else caps
    bufs = [C.create_string_buffer(b, max(len(b), 1)) for b in blocks]
    outs = [np.zeros(max(c, 1), dtype=np.int32) for c in caps]
    inp = (C.c_void_p * n)(*[C.addressof(x) for x in bufs])
    ilen = (C.c_uint32 * n)(*[len(b) for b in blocks])
    outp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    cap = (C.c_uint32 * n)(*caps)
    cnt = (C.c_uint32 * n)(); st = (C.c_int32 * n)()
    f = nat.lib.hg_cram_itf8_decode_host
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(eng._h, inp, ilen, n, outp, cap, cnt, st)
    return rc, [outs[i][:cnt[i]].copy() for i in range(n)], list(st)


def gpu_encode(nat, eng, cols, caps=None):
    n = len(cols)
    cols = [np.ascontiguousarray(c, dtype=np.int32) for c in cols]
    caps = [5 * len(c) + 8 for c in cols] if caps is None else caps
    outs = [C.create_string_buffer(max(c, 1)) for c in caps]
    inp = (C.c_void_p * n)(*[c.ctypes.data if len(c) else 0 for c in cols])
    nv = (C.c_uint32 * n)(*[len(c) for c in cols])
    outp = (C.c_void_p * n)(*[C.addressof(o) for o in outs])
    cap = (C.c_uint32 * n)(*caps)
    ol = (C.c_uint32 * n)(); st = (C.c_int32 * n)()
    f = nat.lib.hg_cram_itf8_encode_host
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(eng._h, inp, nv, n, outp, cap, ol, st)
    return rc, [outs[i].raw[:ol[i]] for i in range(n)], list(st)


@pytest.mark.gpu
def test_gpu_decode_and_encode_equal_oracle(engine):
    from htslib_amd import _native as nat
    L = orc()
    sets = value_sets()
    blocks = [orc_encode(L, v) for v in sets]
    rc, cols, st = gpu_decode(nat, engine, blocks)
    assert rc == 0 and st == [0] * len(sets)
    for v, got in zip(sets, cols):
        assert np.array_equal(got, v), len(v)
    rc, enc, st = gpu_encode(nat, engine, sets)
    assert rc == 0 and st == [0] * len(sets)
    assert enc == blocks
    # arbitrary bytes, and blocks that end inside their last value (the reference's *err): status -1, nothing delivered
    rng = np.random.default_rng(5)
    odd = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (1, 63, 64, 65, 4095, 4096, 4097, 4100, 12289, 100000)]
    odd += [b[:-k] for b in blocks[2:3] + blocks[12:16] for k in (1, 2, 3) if len(b) > 3]
    rc, cols, st = gpu_decode(nat, engine, odd)
    for b, got, s in zip(odd, cols, st):
        r, exp = orc_decode(L, b)
        assert (s != 0) == (r < 0), len(b)
        if r >= 0:
            assert np.array_equal(got, exp), len(b)
        else:
            assert len(got) == 0
    assert rc == (0 if all(s == 0 for s in st) else -6) or rc in (0, -6)
    # no room: one value too few
    rc, cols, st = gpu_decode(nat, engine, [blocks[5], blocks[6]], caps=[len(sets[5]) - 1, len(sets[6])])
    assert st[0] == -1 and st[1] == 0 and np.array_equal(cols[1], sets[6])
    rc, enc2, st = gpu_encode(nat, engine, [sets[13]], caps=[len(blocks[13]) - 1])
    assert st == [-1]


@pytest.mark.gpu
def test_gpu_decodes_fixture_columns_to_the_sam_values(engine):
    from htslib_amd import _native as nat
    cols = fixture_columns()
    rc, got, st = gpu_decode(nat, engine, [b for _, b, _ in cols])
    assert rc == 0 and st == [0] * len(cols)
    for (name, _, vals), g in zip(cols, got):
        assert np.array_equal(g, vals), name


# ---------------------------------------------------------------------------------------------- BYTE_ARRAY_STOP (read names, string tags)
def orc_split(L, b: bytes, stop: int, cap=None):
    cap = len(b) + 2 if cap is None else cap
    off = np.zeros(max(cap, 1), dtype=np.uint32)
    n = L.orc_byte_array_stop_split(b, len(b), stop, off.ctypes.data, cap)
    return n, (off[:n + 1].copy() if n >= 0 else None)


def bas_blocks():
    rng = np.random.default_rng(77)
    names = [b"", b"\t", b"read1\t", b"a\tb\t\tc\t"]
    names.append(b"".join(b"IL%d_%d:%d:%d#0\t" % (rng.integers(1, 9), rng.integers(1, 99), rng.integers(1, 9999), rng.integers(1, 99999)) for _ in range(20000)))
    names.append(b"".join(bytes(rng.integers(65, 91, rng.integers(0, 40), dtype=np.uint8)) + b"\0" for _ in range(5000)))
    names.append(b"\0" * 3000)                                                       # empty items only
    names.append(bytes(rng.integers(1, 256, 70000, dtype=np.uint8)) + b"\0")          # one long item
    r = bytes(rng.integers(0, 4, 100000, dtype=np.uint8))
    names.append(r[:r.rfind(b"\0") + 1])
    stops = [9, 9, 9, 9, 9, 0, 0, 0, 0]
    return names, stops


def test_byte_array_stop_oracle_equals_reference_function(built, tmp_path):
    if not os.path.isdir(REF):
        pytest.skip("no reference checkout here")
    subprocess.run(["bash", os.path.join(ROOT, "tests", "native", "gen_bas_ref.sh"), str(tmp_path)], check=True)
    exe = str(tmp_path / "bas_ref")
    L = orc()
    blocks, stops = bas_blocks()
    for b, st in zip(blocks, stops):
        for cut in (0, 1):
            bb = b[:len(b) - cut] if cut and len(b) > 1 and b[-2] != st else b
            r = subprocess.run([exe, str(st)], input=bb, capture_output=True, check=True).stdout.decode().split("\n")
            end = [x for x in r if x.startswith("END")][0].split()
            sizes = [int(x) for x in r if x and not x.startswith("END")]
            n, off = orc_split(L, bb, st)
            if int(end[2]) != 0:
                assert n == -1, (len(bb), st)
            else:
                assert n == len(sizes) and [int(off[k + 1] - off[k] - 1) for k in range(n)] == sizes, (len(bb), st)


@pytest.mark.gpu
def test_gpu_byte_array_stop_equals_oracle(engine):
    from htslib_amd import _native as nat
    L = orc()
    blocks, stops = bas_blocks()
    blocks = blocks + [blocks[4][:-1], blocks[5][:-3]]                                 # unterminated tails
    stops = stops + [9, 0]
    n = len(blocks)
    caps = [len(b) + 2 for b in blocks]
    bufs = [C.create_string_buffer(b, max(len(b), 1)) for b in blocks]
    outs = [np.zeros(c, dtype=np.uint32) for c in caps]
    inp = (C.c_void_p * n)(*[C.addressof(x) for x in bufs])
    ilen = (C.c_uint32 * n)(*[len(b) for b in blocks])
    stp = (C.c_uint8 * n)(*stops)
    outp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    cap = (C.c_uint32 * n)(*caps)
    cnt = (C.c_uint32 * n)(); st = (C.c_int32 * n)()
    f = nat.lib.hg_cram_byte_array_stop_host
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(engine._h, inp, ilen, stp, n, outp, cap, cnt, st)
    assert rc in (0, -6)
    for i, (b, s_) in enumerate(zip(blocks, stops)):
        en, eoff = orc_split(L, b, s_)
        if en < 0:
            assert st[i] == -1, i
        else:
            assert st[i] == 0 and cnt[i] == en and np.array_equal(outs[i][:en + 1], eoff), i
    # the read names of a reference fixture: items == the .sam twin's QNAMEs
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "rans4x8", "MANIFEST.json")))
    rn = [e for e in man.values() if e["series"] == "RN" and e["expected_hex"]]
    for e in rn:
        b = bytes.fromhex(e["expected_hex"])
        en, eoff = orc_split(L, b, b[-1])
        assert en > 0
