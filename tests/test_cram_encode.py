"""The ENCODE side of the CRAM record layer (htslib_amd/csrc/cram_encode_core.h; reference cram_encode_slice + process_one_read, cram/cram_encode.c:572-793,
1096-1209, 3382-3700): BAM records -> data series blocks + compression / slice headers, as per-record walks and prefix sums.

The bar is the task's for a writer: the output must be valid CRAM that the PINNED decoder (tests/test_cram_records.py: fixtures vs SAM / BAM twins) reads
back field for field.  CPU: the shared per-record source run from plain loops (tests/native/cram_records_host.cpp).  GPU: the kernels of
cram_encode.hip, checked through BAM -> CRAM slices -> BAM byte equality."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from tests import refutil
from tests import test_cram_records as T
from tests.test_cram_records_fast import hostlib, fast_call  # noqa: F401  (fixture)

_vp = C.c_void_p
NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def reg2bin(beg, end):
    end -= 1
    for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh: return off + (beg >> sh)
    return 0


def bam_record(rec, aux=b""):
    """one decoded record [name, flag, ref_id, pos (1-based), mapq, cigar [[len, op]], mate_ref, mate_pos, tlen, seq, qual, ...] -> BAM bytes"""
    name, flag, ref, pos, mapq, cigar, mref, mpos, tlen, seq, qual = rec[:11]
    seq = "" if seq == "*" else seq
    nm = name.encode("latin1") + b"\0"
    rl = sum(l for l, op in cigar if op in (0, 2, 3, 7, 8)) or 1
    packed = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq): packed[i >> 1] |= NT16.get(c, 15) << (4 if i % 2 == 0 else 0)
    q = bytes([0xff] * len(seq)) if qual == "*" else bytes(ord(c) - 33 for c in qual)
    body = struct.pack("<iiBBHHHiiii", ref, pos - 1, len(nm), mapq, reg2bin(pos - 1, pos - 1 + rl) if not flag & 4 else reg2bin(pos - 1, pos), len(cigar), flag, len(seq), mref, mpos - 1, tlen)
    body += nm + b"".join(struct.pack("<I", l << 4 | op) for l, op in cigar) + bytes(packed) + q + aux
    return struct.pack("<i", len(body)) + body


class RefSeq(C.Structure):
    _fields_ = [("bases", _vp), ("len", C.c_uint64)]


def parse_blob(b, nrec, refs, expect):
    p = 0
    def take():
        nonlocal p
        ln = struct.unpack_from("<I", b, p)[0]; p += 4; v = b[p:p + ln]; p += ln; return v
    comp, sh = take(), take()
    nb = struct.unpack_from("<I", b, p)[0]; p += 4
    blocks = []
    for _ in range(nb):
        cid, ln = struct.unpack_from("<iI", b, p); p += 8; blocks.append((cid, b[p:p + ln])); p += ln
    assert p == len(b)
    return {"comp_hdr": comp, "slice_hdr": sh, "core": b"", "blocks": blocks, "nrec": nrec, "refs": refs, "expect": expect}


def encode_host(L, bam, nrec, per_slice, refs, rg_names=(), counter=0):
    """refs: list of bytes / None per reference id -> (status, [blob per slice])"""
    keep = [C.create_string_buffer(r, len(r)) if r is not None else None for r in refs]
    arr = (RefSeq * max(len(refs), 1))(*[RefSeq(C.addressof(k), len(r)) if k is not None else RefSeq(None, 0) for k, r in zip(keep, refs)])
    rg = [r.encode() for r in rg_names]; rgp = (C.c_char_p * max(len(rg), 1))(*rg)
    ns = (nrec + per_slice - 1) // per_slice
    out = np.zeros(len(bam) * 6 + 65536 * ns + 4096, np.uint8); off = np.zeros(ns + 2, np.uint64); st = np.full(ns + 1, 9, np.int32)
    L.hgr_host_encode_slices.restype = C.c_long
    L.hgr_host_encode_slices.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_uint32, _vp, C.c_int, _vp, C.c_int, C.c_int64, _vp, C.c_size_t, _vp, C.c_size_t, _vp]
    n = L.hgr_host_encode_slices(bam, len(bam), nrec, per_slice, C.cast(arr, _vp), len(refs), C.cast(rgp, _vp), len(rg), counter, out.ctypes.data, len(out), off.ctypes.data, ns + 1, st.ctypes.data)
    assert n == ns, n
    return st[:ns], [bytes(out[int(off[i]):int(off[i + 1])]) for i in range(ns)]


def _records_and_bam(hostlib, slices, major, nref, rg_names=()):
    """decode (pinned chain decoder, stored tags only) -> (records per slice, aux bytes per slice, BAM bytes of all records; like cram_to_bam the RG series
    comes out as an RG:Z tag at the end)"""
    T.DECODE_MD[0] = 0
    try:
        st, got = T.decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, major, nref)
        aux = T.decode.last_aux
    finally:
        T.DECODE_MD[0] = -1
    assert (st == 0).all()
    rgs = T.decode.last_rg
    tag = lambda i: b"RGZ" + rg_names[i].encode() + b"\0" if 0 <= i < len(rg_names) else b""
    return got, aux, b"".join(bam_record(r, a + tag(i)) for g, ax, rg in zip(got, aux, rgs) for r, a, i in zip(g, ax, rg))


def test_synthetic_slices_survive_encode_and_decode(hostlib):
    from htslib_amd import synth_cram
    rng = np.random.default_rng(41)
    slices = [synth_cram.make_slice(rng, 1500, 100, tags=True), synth_cram.make_slice(rng, 300, 151, unmapped_every=3), synth_cram.make_slice(rng, 1, 50, ref_len=900),
              synth_cram.make_slice(rng, 257, 64, detached_every=2, tags=True)]
    fc = fast_call(hostlib)
    for s in slices:
        got, aux, bam = _records_and_bam(hostlib, [s], 3, 1)
        ref = s["refs"][0][2]
        for per in (s["nrec"], 100):
            st, blobs = encode_host(hostlib, bam, s["nrec"], per, [ref])
            assert (st == 0).all(), st
            again = []
            for i, b in enumerate(blobs):
                n_i = min(per, s["nrec"] - i * per)
                s2 = parse_blob(b, n_i, s["refs"], got[0][i * per:i * per + n_i])
                T.DECODE_MD[0] = 0
                try:
                    st2, two = T.decode(hostlib.hgr_host_records_bound, fc, [s2], 3, 1)
                finally:
                    T.DECODE_MD[0] = -1
                assert st2[0] == 0 and fc.path[0] == 1                 # what the encoder writes is decoded by the data-parallel passes
                again += two[0]
            assert again == got[0], [(x, y) for x, y in zip(again, got[0]) if x != y][:2]


def test_fixture_records_survive_encode_and_decode(hostlib):
    """every slice of the reference's 34 CRAM fixtures: decoded (pinned), turned into BAM records, encoded, decoded again -- same records.  Refused (-3):
    slices holding a record with a CIGAR but no bases (CF_NO_SEQ) and the 511-tag stress file (more than ENC_MAX_TAGS distinct tags in a slice).
    A stored RG:Z tag (htsjdk keeps it as a tag) moves to the RG series, as in the reference (cram_encode.c:2683-2700)."""
    import json
    rg_of = {f["file"]: [r if isinstance(r, str) else r[0] for r in (f.get("rg") or [])] for f in json.load(open(T.GOLD))}
    done = refused = 0
    for fname, major, nref, s in T.load_slices():
        got, aux, bam = _records_and_bam(hostlib, [s], major, nref, rg_of.get(fname, []))
        rg0 = T.decode.last_rg[0]
        refs = [None] * max(nref, 1)
        for t, a, b, ln in s["refs"]:
            full = bytearray(b"N" * ln); full[a - 1:a - 1 + len(b)] = b; refs[t] = bytes(full)
        names = list(rg_of.get(fname, []))
        for r in got[0]:                                                 # read groups named by a stored RG:Z tag but missing from the header list
            for t in r[11]:
                if t.startswith("RG:Z:") and t[5:] not in names: names.append(t[5:])
        st, blobs = encode_host(hostlib, bam, s["nrec"], max(s["nrec"], 1), refs, names)
        if st[0] == -3:
            ntags = len({t[:4] for r in got[0] for t in r[11]})
            assert any(r[9] == "*" and r[5] for r in got[0]) or ntags > 64, (fname, ntags)
            refused += 1; continue
        assert st[0] == 0, (fname, st)
        s2 = parse_blob(blobs[0], s["nrec"], s["refs"], got[0])
        T.DECODE_MD[0] = 0
        try:
            st2, two = T.decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, [s2], 3, nref)
            rg2 = T.decode.last_rg[0]
        finally:
            T.DECODE_MD[0] = -1
        assert st2[0] == 0, fname
        for x, y, g0, g2 in zip(two[0], got[0], rg0, rg2):
            stored = [t for t in y[11] if t.startswith("RG:Z:")]
            want_rg = names.index(stored[0][5:]) if stored else g0
            assert x[:11] == y[:11] and x[11] == [t for t in y[11] if not t.startswith("RG:Z:")] and g2 == want_rg, (fname, x, y, g0, g2)
        done += len(got[0])
    assert done >= 200 and refused <= 7, (done, refused)


# ---------------------------------------------------------------- on the MI355X: cram_encode.hip ----
def encode_gpu(engine, bam, nrec, per_slice, refs, rg_names=(), counter=0):
    from htslib_amd import _native as nat
    keep = [C.create_string_buffer(r, len(r)) if r is not None else None for r in refs]
    arr = (RefSeq * max(len(refs), 1))(*[RefSeq(C.addressof(k), len(r)) if k is not None else RefSeq(None, 0) for k, r in zip(keep, refs)])
    rg = [r.encode() if isinstance(r, str) else r for r in rg_names]; rgp = (C.c_char_p * max(len(rg), 1))(*rg)
    ns = (nrec + per_slice - 1) // per_slice
    out = np.zeros(len(bam) * 6 + 65536 * ns + 4096, np.uint8); off = np.zeros(ns + 2, np.uint64); st = np.full(ns + 1, 9, np.int32); total = C.c_uint64()
    b = C.create_string_buffer(bam, len(bam))
    rc = nat.lib.hg_cram_encode_slices_host(engine._h, C.cast(b, _vp), len(bam), nrec, per_slice, C.cast(arr, _vp), len(refs), C.cast(rgp, _vp) if rg else None, len(rg), counter,
                                            out.ctypes.data, len(out), off.ctypes.data, ns + 1, st.ctypes.data, C.byref(total))
    assert rc in (0, -6), rc
    return st[:ns], [bytes(out[int(off[i]):int(off[i + 1])]) for i in range(ns)]


@pytest.mark.gpu
def test_gpu_encoder_counts_the_records_itself(engine):
    """hg_cram_encode_slices_host2 with *nrec_io = 0: bam_read1's framing on the device counts the records, the survey pass sums l_seq per slice (the container
    header's base count) -- the same blobs as the call that is told the count, the sums of a host walk; a stream that ends inside a record is refused"""
    from htslib_amd import _native as nat, synth_cram
    rng = np.random.default_rng(47)
    s = synth_cram.make_slice(rng, 2500, 120, tags=True, unmapped_every=7)
    keep = []
    arr = nat.cram_slice_array([s], keep)
    bases = s["nrec"] * 130 + 4096
    bam, rec_off, st = engine.cram_decode_bam(arr, 1, 3, 1, [], bases, bases * 4 + 600 * s["nrec"])
    assert st[0] == 0
    bam = bytes(bam)
    ref = s["refs"][0][2]
    per = 600
    st_g, blobs = encode_gpu(engine, bam, s["nrec"], per, [ref])
    assert (st_g == 0).all()
    kr = C.create_string_buffer(ref, len(ref)); ra = (RefSeq * 1)(RefSeq(C.addressof(kr), len(ref)))
    ns_max = len(bam) // 36 // per + 2
    out = np.zeros(len(bam) * 6 + 65536 * ns_max + 4096, np.uint8); off = np.zeros(ns_max + 1, np.uint64); stt = np.full(ns_max, 9, np.int32); total = C.c_uint64()
    sb = np.zeros(ns_max, np.uint64); n = C.c_size_t(0)
    b = C.create_string_buffer(bam, len(bam))
    f = nat.lib.hg_cram_encode_slices_host2
    f.argtypes = [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint32, _vp, C.c_int, _vp, C.c_int, C.c_int64, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.POINTER(C.c_uint64), _vp]
    rc = f(engine._h, C.cast(b, _vp), len(bam), C.byref(n), per, C.cast(ra, _vp), 1, None, 0, 0, out.ctypes.data, len(out), off.ctypes.data, ns_max, stt.ctypes.data, C.byref(total), sb.ctypes.data)
    ns = (s["nrec"] + per - 1) // per
    assert rc == 0 and n.value == s["nrec"] and (stt[:ns] == 0).all()
    assert [bytes(out[int(off[i]):int(off[i + 1])]) for i in range(ns)] == blobs
    lseq, at = [], 0
    while at < len(bam):
        bs = int.from_bytes(bam[at:at + 4], "little"); lseq.append(int.from_bytes(bam[at + 20:at + 24], "little")); at += 4 + bs
    assert [int(x) for x in sb[:ns]] == [sum(lseq[i * per:(i + 1) * per]) for i in range(ns)]
    n = C.c_size_t(0)
    rc = f(engine._h, C.cast(b, _vp), len(bam) - 7, C.byref(n), per, C.cast(ra, _vp), 1, None, 0, 0, out.ctypes.data, len(out), off.ctypes.data, ns_max, stt.ctypes.data, C.byref(total), sb.ctypes.data)
    assert rc == -1                                                     # HG_EINVAL: the last record is cut short
    n = C.c_size_t(s["nrec"] + 1)
    rc = f(engine._h, C.cast(b, _vp), len(bam), C.byref(n), per, C.cast(ra, _vp), 1, None, 0, 0, out.ctypes.data, len(out), off.ctypes.data, ns_max, stt.ctypes.data, C.byref(total), sb.ctypes.data)
    assert rc == -1                                                     # a count that is not the stream's


@pytest.mark.gpu
def test_gpu_encoder_equals_the_cpu_compile_and_round_trips_to_the_same_bam(engine, hostlib):
    """synthetic slices -> BAM (device: cram_decode_slice + cram_to_bam) -> CRAM slices (device encoder) -> BAM again: byte-identical streams, for whole
    slices and for a different slicing; the encoder's blobs equal the CPU compile's byte for byte"""
    from htslib_amd import _native as nat, synth_cram
    rng = np.random.default_rng(43)
    slices = [synth_cram.make_slice(rng, 3000, 100, tags=True), synth_cram.make_slice(rng, 700, 151, unmapped_every=3), synth_cram.make_slice(rng, 1, 50, ref_len=900),
              synth_cram.make_slice(rng, 513, 64, detached_every=2, tags=True)]
    for s in slices:
        keep = []
        arr = nat.cram_slice_array([s], keep)
        bases = s["nrec"] * 160 + 4096
        bam, rec_off, st = engine.cram_decode_bam(arr, 1, 3, 1, [], bases, bases * 4 + 600 * s["nrec"])
        assert st[0] == 0
        bam = bytes(bam)
        ref = s["refs"][0][2]
        for per in (s["nrec"], 257):
            st_g, blobs = encode_gpu(engine, bam, s["nrec"], per, [ref])
            st_c, blobs_c = encode_host(hostlib, bam, s["nrec"], per, [ref])
            assert (st_g == 0).all() and (st_c == 0).all()
            assert blobs == blobs_c                                     # the same source on both sides: same bytes
            again = [parse_blob(b, min(per, s["nrec"] - i * per), s["refs"], []) for i, b in enumerate(blobs)]
            keep2 = []
            arr2 = nat.cram_slice_array(again, keep2)
            h = engine.cram_batch_stage(arr2, len(again), 3, 1, bases)
            try:
                d, nb, nr, nf, st2 = engine.cram_batch_decode_bam(h, len(again))
                assert (st2 == 0).all() and nf == len(again) and nr == s["nrec"]
                bam2 = bytes(engine.cram_batch_read_bam(h, nb))
            finally:
                engine.cram_batch_free(h)
            assert bam2 == bam


@pytest.mark.gpu
def test_gpu_encoder_on_the_reference_fixtures(engine):
    """the reference's CRAM fixtures: file -> BAM (device) -> CRAM slices (device encoder, one slice per file) -> BAM: the same records (RG:Z included: it
    travels as the RG series).  Files with a CIGAR-without-bases record or the 511-tag stress file are refused with -3."""
    import json
    from htslib_amd import _native as nat
    done = refused = 0
    for f in json.load(open(T.GOLD)):
        slices = [s for fname, major, nref, s in T.load_slices() if fname == f["file"]]
        rg = [r if isinstance(r, str) else r[0] for r in (f.get("rg") or [])]
        keep = []
        arr = nat.cram_slice_array(slices, keep)
        nrec = sum(s["nrec"] for s in slices)
        bases = sum(len(e[9]) for s in slices for e in s["expect"]) + 4096
        bam, rec_off, st = engine.cram_decode_bam(arr, len(slices), f["major"], f["nref"], rg, bases, 1 << 22)
        assert (st == 0).all(), f["file"]
        bam = bytes(bam)
        refs = [None] * max(f["nref"], 1)
        for s in slices:
            for t, a, b, ln in s["refs"]:
                if refs[t] is None: refs[t] = bytearray(b"N" * ln)
                refs[t][a - 1:a - 1 + len(b)] = b
        refs = [bytes(r) if r is not None else None for r in refs]
        st_g, blobs = encode_gpu(engine, bam, nrec, max(nrec, 1), refs, rg)
        if st_g[0] == -3: refused += 1; continue
        assert st_g[0] == 0, (f["file"], st_g)
        allrefs = [x for s in slices for x in s["refs"]]
        s2 = parse_blob(blobs[0], nrec, [(t, 1, r, len(r)) for t, r in enumerate(refs) if r is not None], [])
        keep2 = []
        arr2 = nat.cram_slice_array([s2], keep2)
        bam2, _, st2 = engine.cram_decode_bam(arr2, 1, 3, f["nref"], rg, bases, 1 << 22)
        assert st2[0] == 0, f["file"]
        one, two = T._parse_bam_records(bam), T._parse_bam_records(bytes(bam2))
        assert [r[0] for r in one] == [r[0] for r in two], (f["file"], [(x[0], y[0]) for x, y in zip(one, two) if x[0] != y[0]][:1])
        done += len(one)
    assert done >= 200 and refused <= 6, (done, refused)


def _file_to_bam(engine, cram, seqs, flags=0):
    from htslib_amd import _native as nat
    keep = [C.create_string_buffer(bytes(q), len(q)) if q is not None else None for q in seqs]
    arr = (RefSeq * max(len(seqs), 1))(*[RefSeq(C.addressof(k), len(q)) if k is not None else RefSeq(None, 0) for k, q in zip(keep, seqs)])
    out = np.zeros(max(1 << 22, len(cram) * 40), np.uint8); total = C.c_uint64(); n = C.c_uint64()
    cb = C.create_string_buffer(cram, len(cram))
    rc = nat.lib.hg_cram_file_to_bam_host2(engine._h, C.cast(cb, _vp), len(cram), C.cast(arr, _vp), len(seqs), out.ctypes.data, len(out), C.byref(total), C.byref(n), flags, None)
    return rc, bytes(out[:total.value]), n.value


def _bam_to_cram(engine, bam, seqs, per_slice=0, flags=0, level=5):
    """flags: 1 = CRAM 3.1 (rANS Nx16 + tok3 sets), 3 = + the range coder's sets (hg_bam_to_cram_host2)"""
    from htslib_amd import _native as nat
    keep = [C.create_string_buffer(bytes(q), len(q)) if q is not None else None for q in seqs]
    arr = (RefSeq * max(len(seqs), 1))(*[RefSeq(C.addressof(k), len(q)) if k is not None else RefSeq(None, 0) for k, q in zip(keep, seqs)])
    out = np.zeros(len(bam) * 2 + (1 << 20), np.uint8); total = C.c_uint64(); n = C.c_uint64()
    b = C.create_string_buffer(bam, len(bam))
    rc = nat.lib.hg_bam_to_cram_host2(engine._h, C.cast(b, _vp), len(bam), C.cast(arr, _vp), len(seqs), per_slice, level, flags, out.ctypes.data, len(out), C.byref(total), C.byref(n))
    return rc, bytes(out[:total.value]), n.value


@pytest.mark.gpu
def test_gpu_bam_to_cram_file_and_back(engine):
    """whole files: the reference's CRAM fixtures -> BAM stream (hg_cram_file_to_bam_host) -> CRAM 3.0 file (hg_bam_to_cram_host: encoder, block auto-tuner
    with gzip / rANS 4x8, container framing, EOF container) -> BAM stream again: identical bytes, header included.  Then a 150 bp synthetic BAM of 20 000
    records in slices of 3 000 without a reference (every base stored)."""
    import json
    from htslib_amd import synth
    ok = refused = regenerated = 0
    for f in json.load(open(T.GOLD)):
        cram = T.unpack(f["cram"])
        spans = {}
        for s in f["slices"]:
            for t, a, b, ln in s["refs"]:
                spans.setdefault(t, (ln, []))[1].append((a, T.unpack(b)))
        seqs = []
        for i, name in enumerate(f["ref_names"]):
            if f["full_refs"]: seqs.append(bytearray(T.unpack(dict(f["full_refs"])[name])))
            elif i in spans:
                sq = bytearray(b"N" * spans[i][0])
                for a, b in spans[i][1]: sq[a - 1:a - 1 + len(b)] = b
                seqs.append(sq)
            else: seqs.append(None)
        rc, bam, n = _file_to_bam(engine, cram, seqs)
        assert rc == 0, f["file"]
        rc2, cram2, n2 = _bam_to_cram(engine, bam, seqs)
        if rc2 == -6: refused += 1; continue                            # a slice the encoder does not cover (CF_NO_SEQ, 511 distinct tags)
        assert rc2 == 0 and n2 == n and cram2[:6] == b"CRAM\x03\x00" and cram2[-38:-34] == b"\x0f\x00\x00\x00", (f["file"], rc2)
        rc3, bam3, n3 = _file_to_bam(engine, cram2, seqs)
        assert rc3 == 0 and n3 == n, (f["file"], rc3)
        if bam3 != bam:
            # a file written WITHOUT a reference (RR = 0: bases stored, no MD / NM on decoding) comes back from our writer as a file WITH one (RR = 1):
            # the second decoding regenerates MD:Z / NM.  Everything else must agree.
            import struct
            hb = 12 + struct.unpack_from("<i", bam, 4)[0]
            for _ in range(struct.unpack_from("<i", bam, hb - 4)[0]): hb += 8 + struct.unpack_from("<i", bam, hb)[0]
            assert bam3[:hb] == bam[:hb], f["file"]
            one, two = T._parse_bam_records(bam[hb:]), T._parse_bam_records(bam3[hb:])
            strip = lambda tags: [t for t in tags if not t.startswith(("MD:Z:", "NM:"))]
            assert len(one) == len(two) and all(x[0][:11] == y[0][:11] and strip(x[0][11]) == strip(y[0][11]) for x, y in zip(one, two)), f["file"]
            regenerated += 1
        ok += 1
    assert ok >= 28 and refused <= 6 and regenerated < ok, (ok, refused, regenerated)
    plain, _, _ = synth.bam_stream(6 << 20, 0x5EED0001, 0, True)
    # cut at a record boundary: walk the records of the stream
    import struct
    lt = struct.unpack_from("<i", plain, 4)[0]; p = 8 + lt; nref = struct.unpack_from("<i", plain, p)[0]; p += 4
    for _ in range(nref): p += 4 + struct.unpack_from("<i", plain, p)[0] + 4
    q, nrec = p, 0
    while q + 4 <= len(plain) and nrec < 20000:
        nxt = q + 4 + struct.unpack_from("<i", plain, q)[0]
        if nxt > len(plain): break
        q = nxt; nrec += 1
    bam = plain[:q]
    rc, cram, n = _bam_to_cram(engine, bam, [None] * nref, 3000)
    assert rc == 0 and n == nrec, rc
    rc, back, n2 = _file_to_bam(engine, cram, [None] * nref)
    assert rc == 0 and n2 == nrec
    one, two = T._parse_bam_records(bam[p:]), T._parse_bam_records(back[p:])
    assert back[:p] == bam[:p] and len(one) == len(two) == nrec
    for x, y in zip(one, two):
        assert x[0][:11] == y[0][:11] and sorted(x[0][11]) == sorted(y[0][11]), (x[0], y[0])   # RG:Z moves to the end of the tag list
    print("synthetic BAM %d bytes -> CRAM %d bytes (%.2fx)" % (len(bam), len(cram), len(bam) / len(cram)))


def _synthetic_bam(nrec_max=40000):
    import struct
    from htslib_amd import synth
    plain, _, _ = synth.bam_stream(12 << 20, 0x5EED0001, 0, True)
    lt = struct.unpack_from("<i", plain, 4)[0]; p = 8 + lt; nref = struct.unpack_from("<i", plain, p)[0]; p += 4
    for _ in range(nref): p += 4 + struct.unpack_from("<i", plain, p)[0] + 4
    q, nrec = p, 0
    while q + 4 <= len(plain) and nrec < nrec_max:
        nxt = q + 4 + struct.unpack_from("<i", plain, q)[0]
        if nxt > len(plain): break
        q = nxt; nrec += 1
    return plain[:q], nref, nrec


@pytest.mark.gpu
def test_gpu_file_decode_sharded_over_several_contexts_is_byte_identical(engine, tmp_path):
    """HTS_GPU_DEVICES names several devices: a whole-file decode cuts its blocks and slices into one contiguous range per entry, each range on its
    own context / host thread, and joins the outputs in file order.  "0,0,0" = three contexts on the one GPU of the test box: the
    sharded result must be the unsharded one, byte for byte.  (The variable is read once per process: the sharded run is a child process.)"""
    import subprocess, sys, os
    bam, nref, nrec = _synthetic_bam()
    rc, cram, n = _bam_to_cram(engine, bam, [None] * nref, 2500)
    assert rc == 0 and n == nrec and nrec >= 20000
    rc, want, n1 = _file_to_bam(engine, cram, [None] * nref)
    assert rc == 0 and n1 == nrec
    (tmp_path / "in.cram").write_bytes(cram)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from htslib_amd import _native as nat\n"
            "from tests.test_cram_encode import _file_to_bam\n"
            "eng = nat.Engine(0)\n"
            "rc, out, n = _file_to_bam(eng, open(%r, 'rb').read(), [None] * %d)\n"
            "assert rc == 0, rc\n"
            "open(%r, 'wb').write(out); print('sharded ok', n)\n") % (refutil.ROOT, str(tmp_path / "in.cram"), nref, str(tmp_path / "out.bam"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HTS_GPU_DEVICES="0,0,0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded ok %d" % nrec in r.stdout, r.stderr[-3000:]
    assert (tmp_path / "out.bam").read_bytes() == want
    # the writer, sharded the same way (one set of cram_metrics per range: a block's method may differ, the records may not)
    (tmp_path / "in.bam").write_bytes(bam)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from htslib_amd import _native as nat\n"
            "from tests.test_cram_encode import _bam_to_cram\n"
            "eng = nat.Engine(0)\n"
            "rc, out, n = _bam_to_cram(eng, open(%r, 'rb').read(), [None] * %d, 2500)\n"
            "assert rc == 0, rc\n"
            "open(%r, 'wb').write(out); print('sharded ok', n)\n") % (refutil.ROOT, str(tmp_path / "in.bam"), nref, str(tmp_path / "out.cram"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HTS_GPU_DEVICES="0,0,0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded ok %d" % nrec in r.stdout, r.stderr[-3000:]
    rc, again, n2 = _file_to_bam(engine, (tmp_path / "out.cram").read_bytes(), [None] * nref)
    assert rc == 0 and n2 == nrec and again == want
