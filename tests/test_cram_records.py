"""CRAM record decoding (the record loop of cram_decode_slice, cram/cram_decode.c:2346-3026, with cram_decode_seq's
feature walk and cram_decode_slice_xref) -- PINNED on the reference's own fixtures: every slice of the 34 CRAM v3.0 files that have a
SAM / BAM twin (tests/golden/cram_records.json, frozen by make_golden_cram_records.py with NO CRAM decoding code involved on the
expectation side) must decode to the twin's QNAME, FLAG, RNAME, POS, MAPQ, CIGAR, RNEXT, PNEXT, TLEN, SEQ, QUAL (the bases are rebuilt
from the reference's .fa files plus the stored edits, cram_decode_seq) and optional tags (cram_decode_aux).  The 31 test/tlen pairs were
written by the reference's authors to pin the mate / template-length logic.
CPU part: the decoder source (htslib_amd/csrc/cram_records_core.h) compiled for the host by tests/native/cram_records_host.cpp.
GPU part: the same fixtures through hg_cram_decode_records_host (one wavefront per slice), plus a replicated batch."""
import base64, ctypes as C, json, os, subprocess, zlib
from struct import error as struct_error

import numpy as np
import pytest

from tests.golden import make_golden_cram_records as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "cram_records.json")
_vp = C.c_void_p
DECODE_MD = [-1]                            # fd->decode_md of the slices built below (hts_open's default; tests that compare stored tags only set 0)


class SliceIn(C.Structure):                 # = hg_cram_slice_blocks
    _fields_ = [("comp_hdr", _vp), ("comp_hdr_len", C.c_uint32), ("slice_hdr", _vp), ("slice_hdr_len", C.c_uint32), ("core", _vp), ("core_len", C.c_uint32),
                ("nblocks", C.c_uint32), ("content_id", _vp), ("data", _vp), ("len", _vp), ("nrefs", C.c_uint32), ("refs", _vp), ("decode_md", C.c_int32)]


class RefIn(C.Structure):                   # = hg_cram_ref_span
    _fields_ = [("ref_id", C.c_int32), ("start", C.c_int64), ("bases", _vp), ("len", C.c_uint32), ("sq_len", C.c_int64)]


class Cols(C.Structure):                    # = hg_cram_record_cols
    _fields_ = [(k, _vp) for k in ("flags", "cram_flags", "ref_id", "len", "rg", "mqual", "mate_ref_id", "ncigar", "name_len", "apos", "aend", "mate_pos", "tlen",
                                   "cigar_off", "name_off", "cigar", "names", "seq_off", "seq", "qual", "aux_off", "aux_len", "aux")]


def unpack(s):
    return zlib.decompress(base64.b64decode(s))


def load_slices():
    """-> [(file, major, nref, slice dict with decoded bytes)]"""
    out = []
    for f in json.load(open(GOLD)):
        for s in f["slices"]:
            out.append((f["file"], f["major"], f["nref"], {"comp_hdr": unpack(s["comp_hdr"]), "slice_hdr": unpack(s["slice_hdr"]), "core": unpack(s["core"]),
                                                          "blocks": [(cid, unpack(d)) for cid, d in s["blocks"]], "nrec": s["nrec"], "expect": s["expect"],
                                                          "refs": [(t, a, unpack(b), ln) for t, a, b, ln in s["refs"]]}))
    return out


def decode(call_bound, call_decode, slices, major, nref, with_seq=True):
    """slices: dicts as above -> (status, per-slice list of record tuples in the twin's layout)"""
    n = len(slices)
    keep, arr = [], (SliceIn * n)()
    same = {}
    for i, s in enumerate(slices):
        ch = same.setdefault(s["comp_hdr"], C.create_string_buffer(s["comp_hdr"], len(s["comp_hdr"])))     # one container -> one buffer
        sh = C.create_string_buffer(s["slice_hdr"], len(s["slice_hdr"])); co = C.create_string_buffer(s["core"], max(len(s["core"]), 1))
        bl = [C.create_string_buffer(d, max(len(d), 1)) for _, d in s["blocks"]]
        ids = np.array([cid for cid, _ in s["blocks"]], dtype=np.int32); lens = np.array([len(d) for _, d in s["blocks"]], dtype=np.uint32)
        ptrs = (_vp * max(len(bl), 1))(*[C.addressof(x) for x in bl])
        rb = [C.create_string_buffer(b, max(len(b), 1)) for _, _, b, _ in s.get("refs", [])]
        ra = (RefIn * max(len(rb), 1))(*[RefIn(t, a, C.addressof(buf), len(b), ln) for (t, a, b, ln), buf in zip(s.get("refs", []), rb)])
        keep.append((ch, sh, co, bl, ids, lens, ptrs, rb, ra))
        arr[i] = SliceIn(C.addressof(ch), len(s["comp_hdr"]), C.addressof(sh), len(s["slice_hdr"]), C.addressof(co), len(s["core"]), len(bl), ids.ctypes.data,
                         C.addressof(ptrs), lens.ctypes.data, len(rb) if with_seq else 0, C.addressof(ra), DECODE_MD[0])
    nrec, ccap, ncap, acap = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert call_bound(n, arr, major, C.byref(nrec), C.byref(ccap), C.byref(ncap), C.byref(acap)) == 0
    R = max(nrec.value, 1)
    i32 = {k: np.full(R, -99, np.int32) for k in ("flags", "cram_flags", "ref_id", "len", "rg", "mqual", "mate_ref_id", "ncigar", "name_len")}
    i64 = {k: np.full(R, -99, np.int64) for k in ("apos", "aend", "mate_pos", "tlen")}
    u64 = {k: np.zeros(R, np.uint64) for k in ("cigar_off", "name_off")}
    cigar = np.zeros(max(ccap.value, 1), np.uint32); names = np.zeros(max(ncap.value, 1), np.uint8)
    seq_cap = sum(len(e[9]) if len(e) > 9 and e[9] != "*" else 0 for s in slices for e in s["expect"]) + 4096     # the containers' base count would do
    seq_off = np.zeros(R, np.uint64); seq = np.zeros(seq_cap, np.uint8); qual = np.zeros(seq_cap, np.uint8)
    aux_off = np.zeros(R, np.uint64); aux_len = np.zeros(R, np.int32); aux = np.zeros(max(acap.value, 1), np.uint8)
    cols = Cols(*[a.ctypes.data for a in list(i32.values()) + list(i64.values()) + list(u64.values()) + [cigar, names]],
                *([seq_off.ctypes.data, seq.ctypes.data, qual.ctypes.data] if with_seq else [None, None, None]),
                *([aux_off.ctypes.data, aux_len.ctypes.data, aux.ctypes.data] if with_seq else [None, None, None]))
    rec_off = np.zeros(n + 1, np.uint64); status = np.full(n, 77, np.int32)
    rc = call_decode(n, arr, major, nref, R, len(cigar), len(names), seq_cap, len(aux), C.byref(cols), rec_off.ctypes.data, status.ctypes.data)
    assert rc in (0, -6), rc                      # HG_EBLOCK: some slice has a non-zero status
    out = []
    for i in range(n):
        recs = []
        if status[i] != 0:                   # a failed slice leaves its columns undefined
            out.append(recs); continue
        for r in range(int(rec_off[i]), int(rec_off[i + 1])):
            co, nc = int(u64["cigar_off"][r]), int(i32["ncigar"][r])
            cg = [[int(c >> 4), int(c & 15)] for c in cigar[co:co + nc]]
            no, nl = int(u64["name_off"][r]), int(i32["name_len"][r])
            recs.append([bytes(names[no:no + nl]).decode("latin1"), int(i32["flags"][r]), int(i32["ref_id"][r]), int(i64["apos"][r]), int(i32["mqual"][r]), cg,
                         int(i32["mate_ref_id"][r]), int(i64["mate_pos"][r]), int(i64["tlen"][r])])
            if with_seq:
                so, ln = int(seq_off[r]), int(i32["len"][r])
                q = qual[so:so + ln]
                recs[-1] += [bytes(seq[so:so + ln]).decode("latin1") if ln else "*",
                             "*" if ln == 0 or (q == 255).all() else bytes((q + 33).astype(np.uint8)).decode("latin1"),
                             [G.short_tag(t) for t in G.aux_to_text(bytes(aux[int(aux_off[r]):int(aux_off[r]) + int(aux_len[r])]))]]
        out.append(recs)
    decode.last_aend = [[int(i64["aend"][r]) for r in range(int(rec_off[i]), int(rec_off[i + 1]))] for i in range(n)]    # for the index test
    decode.last_aux = [[bytes(aux[int(aux_off[r]):int(aux_off[r]) + int(aux_len[r])]) for r in range(int(rec_off[i]), int(rec_off[i + 1]))] if with_seq and status[i] == 0 else [] for i in range(n)]
    decode.last_rg = [[int(i32["rg"][r]) for r in range(int(rec_off[i]), int(rec_off[i + 1]))] for i in range(n)]
    return status, out


def check_against_twin(fname, got, expect):
    assert len(got) == len(expect), fname
    for g, e in zip(got, expect):
        g, e = list(g), list(e)
        if e[1] & 4:                         # unmapped: CRAM does not store a mapping quality or a CIGAR for these
            e[4] = 0; e[5] = []
        if len(g) == 9: e = e[:9]            # decoded without bases / qualities / tags
        else:
            # tags: what the CRAM stores must be in the twin with the same value; the writer drops RG (kept as the RG series; cram_to_bam puts it back)
            stored, twin = list(g[11]), e[11]
            twin = [t[:5] + t[5:].upper() if t[2:5] == ":H:" else t for t in twin]
            hexed = {t[:2]: t for t in twin if t[2:5] == ":H:"}          # htsjdk stores an H (hex string) tag as a B:c array of its bytes
            for k, t in enumerate(stored):
                if t[2:6] == ":B:c" and t[:2] in hexed:
                    vals = [int(v) & 0xFF for v in t[7:].split(",")] if len(t) > 6 else []
                    stored[k] = "%s:H:%s" % (t[:2], "".join("%02X" % v for v in vals))
            def canon(t):                                                 # htsjdk has no unsigned types: B:C / B:S / B:I arrays come back as c / s / i with the same bytes
                if t[2:5] != ":B:" or t[5] == "f": return t
                w = {"c": 8, "s": 16, "i": 32}[t[5].lower()]
                return "%s:B:%d%s" % (t[:2], w, "".join(",%d" % (int(v) & ((1 << w) - 1)) for v in t[7:].split(",") if v))
            stored, twin = [canon(t) for t in stored], [canon(t) for t in twin]
            # MD / NM are regenerated from the reference (decode_md): where the twin has them they must agree; a twin without them (the aligner
            # wrote none) does not constrain ours
            twin_has = {t[:2] for t in twin}
            assert all(t in twin or (t[:2] in ("MD", "NM") and t[:2] not in twin_has) for t in stored), (fname, g[0], [t for t in stored if t not in twin])
            assert all(t in stored or t[:2] == "RG" for t in twin), (fname, g[0], [t for t in twin if t not in stored])
            g, e = g[:11], e[:11]
        assert g == e, (fname, g, e)


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("cramrec") / "libcram_records_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"] + os.environ.get("HG_TEST_HOSTLIB_FLAGS", "").split() + ["-o", so, os.path.join(ROOT, "tests", "native", "cram_records_host.cpp")], check=True)
    L = C.CDLL(so)
    L.hgr_host_records_bound.argtypes = [C.c_size_t, _vp, C.c_int, _vp, _vp, _vp, _vp]
    L.hgr_host_decode_records.argtypes = [C.c_size_t, _vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp]
    return L


def test_decoder_source_on_the_cpu_matches_the_sam_twins(hostlib):
    files = {}
    for fname, major, nref, s in load_slices():
        files.setdefault((fname, major, nref), []).append(s)
    assert len(files) == 34
    nrec = 0
    for (fname, major, nref), slices in files.items():
        st, got = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, major, nref)
        assert (st == 0).all(), (fname, st)
        for s, g in zip(slices, got):
            check_against_twin(fname, g, s["expect"]); nrec += len(g)
    assert nrec == 230


def _gpu_calls(engine):
    from htslib_amd import _native as nat
    bound = lambda n, arr, major, a, b, c, d: nat.lib.hg_cram_records_bound(n, C.cast(arr, _vp), major, C.cast(a, _vp), C.cast(b, _vp), C.cast(c, _vp), C.cast(d, _vp))
    def dec(n, arr, major, nref, R, cc, nc, sc, ac, cols, ro, st):
        used = np.zeros(4, np.uint64)
        rc = nat.lib.hg_cram_decode_records_host(engine._h, n, C.cast(arr, _vp), major, nref, R, cc, nc, sc, ac, C.cast(cols, _vp), ro, st, used.ctypes.data)
        _gpu_calls.last_used = used
        return rc
    return bound, dec


@pytest.mark.gpu
def test_gpu_decoder_matches_the_sam_twins(engine):
    bound, dec = _gpu_calls(engine)
    files = {}
    for fname, major, nref, s in load_slices():
        files.setdefault((fname, major, nref), []).append(s)
    nrec = 0
    for (fname, major, nref), slices in files.items():
        st, got = decode(bound, dec, slices, major, nref)
        assert (st == 0).all(), (fname, st)
        for s, g in zip(slices, got):
            check_against_twin(fname, g, s["expect"]); nrec += len(g)
    assert nrec == 230


@pytest.mark.gpu
def test_gpu_decoder_batch_of_slices_and_error_statuses(engine, hostlib):
    """Many slices in one call (all fixtures with one reference, replicated), and damaged slices: the device reports exactly what the
    same source reports on the CPU."""
    bound, dec = _gpu_calls(engine)
    base = [s for fname, major, nref, s in load_slices() if nref == 1]
    slices = [base[i % len(base)] for i in range(600)]
    st, got = decode(bound, dec, slices, 3, 1)
    assert (st == 0).all()
    for s, g in zip(slices, got):
        check_against_twin("batch", g, s["expect"])
    rng = np.random.default_rng(3)
    bad = []
    for i in range(60):
        s = dict(base[i % len(base)])
        kind = i % 4
        if kind == 0: s["core"] = s["core"][:len(s["core"]) // 2]
        elif kind == 1 and s["blocks"]: s["blocks"] = [(cid, d[:len(d) // 2]) for cid, d in s["blocks"]]
        elif kind == 2 and s["blocks"]: s["blocks"] = s["blocks"][1:]
        else:
            c = bytearray(s["core"])
            for _ in range(3):
                if c: c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
            s["core"] = bytes(c)
        bad.append(s)
    st_g, got_g = decode(bound, dec, bad, 3, 1)
    st_c, got_c = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, bad, 3, 1)
    assert (st_g == st_c).all() and (st_g != 0).any()
    for k in range(len(bad)):
        if st_g[k] == 0: assert got_g[k] == got_c[k], k


def crai_of(slices_meta, decoded, major):
    """.crai text of a file from its slices' headers and the decoded ref_id / apos / aend (hg_cram_crai_slice = cram_index_slice)"""
    from htslib_amd import _native as nat
    text = b""
    for meta, recs, ends in zip(slices_meta, decoded, decode.last_aend):
        rid = np.array([r[2] for r in recs], np.int32); ap = np.array([r[3] for r in recs], np.int64)
        ae = np.array(ends, np.int64)                                  # the decoder's alignment ends (cram_record.aend)
        buf = C.create_string_buffer(4096)
        sh = meta["slice_hdr"]
        n = nat.lib.hg_cram_crai_slice(C.cast(C.c_char_p(sh), _vp), len(sh), major, rid.ctypes.data, ap.ctypes.data, ae.ctypes.data, meta["cpos"], meta["landmark"],
                                       meta["slice_bytes"], buf, 4096)
        assert n > 0, n
        text += buf.raw[:n]
    return text.decode()


def test_crai_of_a_multi_reference_file_matches_the_reference_index(hostlib, built):
    """test/range.cram.crai was written by the reference: three multi-reference slices, six index lines whose starts and spans come from
    the decoded alignment positions and ends (cram_index_build_multiref, cram_index.c:632-690)."""
    gold = [f for f in json.load(open(GOLD)) if f["crai"]]
    assert len(gold) == 1 and gold[0]["file"] == "test/range.cram"
    f = gold[0]
    slices = [s for fname, major, nref, s in load_slices() if fname == f["file"]]
    st, got = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, f["major"], f["nref"])
    assert (st == 0).all()
    meta = [{"slice_hdr": unpack(s["slice_hdr"]), "cpos": s["cpos"], "landmark": s["landmark"], "slice_bytes": s["slice_bytes"]} for s in f["slices"]]
    assert crai_of(meta, got, f["major"]) == f["crai"]


@pytest.mark.gpu
def test_gpu_outputs_come_back_packed_and_small_arrays_are_reported(engine):
    from htslib_amd import _native as nat
    bound, dec = _gpu_calls(engine)
    base = [s for fname, major, nref, s in load_slices() if fname == "test/range.cram"]
    slices = [base[i % 3] for i in range(30)]
    st, got = decode(bound, dec, slices, 3, 7)
    assert (st == 0).all()
    used = _gpu_calls.last_used
    ncig = sum(len(r[5]) for g in got for r in g); nname = sum(len(r[0]) for g in got for r in g)
    assert used[0] == ncig and used[1] == nname and used[3] == sum(len(r[9]) for g in got for r in g)       # exactly what the records hold: no gaps
    # the same call with arrays of half the needed size: HG_ENOMEM (-2) and the needed sizes
    tight = lambda n, arr, major, nref, R, cc, nc, sc, ac, cols, ro, stt: dec(n, arr, major, nref, R, int(used[0]) // 2, nc, sc, ac, cols, ro, stt)
    with pytest.raises(AssertionError):
        decode(bound, tight, slices, 3, 7)
    assert _gpu_calls.last_used[0] == ncig


def _check_truth(slices, got):
    for s, g in zip(slices, got):
        assert len(g) == s["nrec"]
        for r, t in zip(g, s["truth"]):
            qual = "*" if not len(t["qual"]) else bytes(q + 33 for q in t["qual"]).decode("latin1")
            assert (r[0], r[1] & ~0x28, r[3], r[5], r[9], r[10]) == (t["name"].decode(), t["flag"], t["pos"], t["cigar"], t["seq"].decode(), qual), (r, t)
            if DECODE_MD[0] == 0: assert r[11] == [G.short_tag(x) for x in G.aux_to_text(t["aux"])], (r[11], t["aux"])     # stored tags only


def test_synthetic_slices_of_production_size_on_the_cpu_compile(hostlib):
    """EXTERNAL-only slices as current htslib writes them (htslib_amd/synth_cram.py), 2 000 records each with clips, substitutions, insertions,
    deletions, unmapped reads, in-slice and detached mates: names, flags, positions, CIGARs, bases and qualities equal what the generator
    put in."""
    from htslib_amd import synth_cram as cram_synth
    rng = np.random.default_rng(21)
    slices = [cram_synth.make_slice(rng, 2000, 100), cram_synth.make_slice(rng, 700, 151, unmapped_every=5), cram_synth.make_slice(rng, 1, 50, ref_len=2000)]
    st, got = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 1)
    assert (st == 0).all(), st
    _check_truth(slices, got)
    tagged = [cram_synth.make_slice(rng, 900, 100, tags=True), cram_synth.make_slice(rng, 40, 151, unmapped_every=3, tags=True)]
    DECODE_MD[0] = 0                                                  # compare the STORED tags: no MD / NM regeneration
    try:
        st, got = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, tagged, 3, 1)
        assert (st == 0).all(), st
        _check_truth(tagged, got)
    finally:
        DECODE_MD[0] = -1


@pytest.mark.gpu
def test_gpu_synthetic_slices_match_the_cpu_compile(engine, hostlib):
    from htslib_amd import synth_cram as cram_synth
    bound, dec = _gpu_calls(engine)
    rng = np.random.default_rng(22)
    slices = [cram_synth.make_slice(rng, int(n), 100) for n in (3000, 1, 2, 63, 64, 65, 1500, 10)] + [cram_synth.make_slice(rng, 800, 151, unmapped_every=4)]
    for mode in ("wave", "lane"):
        os.environ["HG_CRAM_RECORDS_MODE"] = mode
        st, got = decode(bound, dec, slices, 3, 1)
        assert (st == 0).all(), (mode, st)
        _check_truth(slices, got)
        st_c, got_c = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 1)
        assert got == got_c, mode
    del os.environ["HG_CRAM_RECORDS_MODE"]


def _slice_array(slices, keep):
    """ctypes view of slice dicts (as decode() builds it)"""
    n = len(slices)
    arr = (SliceIn * n)()
    same = {}
    for i, s in enumerate(slices):
        ch = same.setdefault(s["comp_hdr"], C.create_string_buffer(s["comp_hdr"], len(s["comp_hdr"])))
        sh = C.create_string_buffer(s["slice_hdr"], len(s["slice_hdr"])); co = C.create_string_buffer(s["core"], max(len(s["core"]), 1))
        bl = [C.create_string_buffer(d, max(len(d), 1)) for _, d in s["blocks"]]
        ids = np.array([cid for cid, _ in s["blocks"]], dtype=np.int32); lens = np.array([len(d) for _, d in s["blocks"]], dtype=np.uint32)
        ptrs = (_vp * max(len(bl), 1))(*[C.addressof(x) for x in bl])
        rb = [C.create_string_buffer(b, max(len(b), 1)) for _, _, b, _ in s.get("refs", [])]
        ra = (RefIn * max(len(rb), 1))(*[RefIn(t, a, C.addressof(buf), len(b), ln) for (t, a, b, ln), buf in zip(s.get("refs", []), rb)])
        keep.append((ch, sh, co, bl, ids, lens, ptrs, rb, ra))
        arr[i] = SliceIn(C.addressof(ch), len(s["comp_hdr"]), C.addressof(sh), len(s["slice_hdr"]), C.addressof(co), len(s["core"]), len(bl), ids.ctypes.data,
                         C.addressof(ptrs), lens.ctypes.data, len(rb), C.addressof(ra), DECODE_MD[0])
    return arr


def _parse_bam_records(b):
    """uncompressed BAM records -> [(fields as in the twin layout ..., bin, raw bytes from refID up to the tags)]"""
    import struct
    out, p = [], 0
    while p < len(b):
        bs = struct.unpack_from("<i", b, p)[0]; q = p + 4; p = q + bs
        tid, pos, lname, mapq, bn, ncig, flag, lseq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", b, q)
        name = b[q + 32:q + 32 + lname - 1].decode("latin1")
        assert b[q + 32 + lname - 1] == 0
        cig = struct.unpack_from("<%dI" % ncig, b, q + 32 + lname)
        a = q + 32 + lname + 4 * ncig
        packed = b[a:a + (lseq + 1) // 2]
        seq = "".join("=ACMGRSVTWYHKDBN"[(packed[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(lseq)) or "*"
        ql = b[a + (lseq + 1) // 2:a + (lseq + 1) // 2 + lseq]
        qual = "*" if lseq == 0 or all(c == 0xFF for c in ql) else bytes(c + 33 for c in ql).decode("latin1")
        tags = [G.short_tag(t) for t in G.aux_to_text(b[a + (lseq + 1) // 2 + lseq:p])]
        out.append(([name, flag, tid, pos + 1, mapq, [[c >> 4, c & 15] for c in cig], mtid, mpos + 1, tlen, seq, qual, tags], bn, bytes(b[q:a + (lseq + 1) // 2 + lseq])))
    return out


@pytest.mark.gpu
def test_gpu_cram_to_bam_records(engine):
    """hg_cram_decode_bam_host: the fixtures' slices come back as uncompressed BAM records whose fields are the twin's; for range.cram the
    bytes from refID up to the tags are the bytes of the reference's own range.bam (bin, l_read_name, packed bases, ... included), and
    the read group comes back as an RG:Z tag."""
    import base64
    from htslib_amd import _native as nat
    gold = json.load(open(GOLD))
    by_file = {}
    for fname, major, nref, s in load_slices():
        by_file.setdefault(fname, []).append(s)
    nrec = 0
    for f in gold:
        slices = by_file[f["file"]]
        keep = []
        arr = _slice_array(slices, keep)
        rg = [r.encode() for r in f["rg"]]
        rgp = (C.c_char_p * max(len(rg), 1))(*rg) if rg else None
        bases = sum(len(e[9]) for s in slices for e in s["expect"] if e[9] != "*") + 64
        n = len(slices)
        out = np.zeros(1 << 22, np.uint8); rec_off = np.zeros(n + 1, np.uint64); total = C.c_uint64(); st = np.full(n, 9, np.int32)
        nr = sum(s["nrec"] for s in slices)
        boff = np.zeros(nr + 1, np.uint64)
        rc = nat.lib.hg_cram_decode_bam_host(engine._h, n, C.cast(arr, _vp), f["major"], f["nref"], C.cast(rgp, _vp) if rg else None, len(rg), bases, out.ctypes.data, len(out),
                                             rec_off.ctypes.data, boff.ctypes.data, C.byref(total), st.ctypes.data)
        assert rc == 0 and (st == 0).all(), (f["file"], rc, st)
        recs = _parse_bam_records(bytes(out[:total.value]))
        expect = [e for s in slices for e in s["expect"]]
        assert len(recs) == len(expect) == nr and int(boff[nr]) == total.value
        for (g, bn, raw), e in zip(recs, expect):
            check_against_twin(f["file"], [g], [e])
            assert [t for t in g[11] if t.startswith("RG:Z:")] == [t for t in e[11] if t.startswith("RG:Z:")], g      # RG:Z comes back from the read-group series
            if len(e) > 12: assert raw == base64.b64decode(e[12]), (f["file"], g[0])       # byte-identical to the reference's BAM record up to the tags
            nrec += 1
    assert nrec == 230
    # a buffer that is too small: HG_ENOMEM and the size needed
    small = np.zeros(64, np.uint8)
    rc = nat.lib.hg_cram_decode_bam_host(engine._h, n, C.cast(arr, _vp), 3, f["nref"], None, 0, bases, small.ctypes.data, len(small), rec_off.ctypes.data, None, C.byref(total), st.ctypes.data)
    assert rc == -3 and total.value > 64


def test_crai_slice_lines_single_reference_and_unsorted(built):
    """cram_index_slice: a single-reference slice is indexed from its header alone; a multi-reference slice whose records go backwards on
    one reference is refused with -2 like the reference ("CRAM file is not sorted by chromosome / position")."""
    from htslib_amd import _native as nat
    from htslib_amd.synth_cram import put_itf8
    buf = C.create_string_buffer(256)
    hdr = put_itf8(3) + put_itf8(1000) + put_itf8(250) + put_itf8(7) + bytes([0]) + put_itf8(5) + put_itf8(0) + put_itf8(-1) + bytes(16)
    n = nat.lib.hg_cram_crai_slice(C.cast(C.c_char_p(hdr), _vp), len(hdr), 3, None, None, None, 4096, 17, 999, buf, 256)
    assert buf.raw[:n] == b"3\t1000\t250\t4096\t17\t999\n"
    multi = put_itf8(-2) + put_itf8(0) + put_itf8(0) + put_itf8(4) + bytes([0]) + put_itf8(5) + put_itf8(0) + put_itf8(-1) + bytes(16)
    rid = np.array([0, 0, 1, -1], np.int32); ap = np.array([10, 20, 5, 0], np.int64); ae = np.array([59, 40, 104, 0], np.int64)
    call = lambda r, a, e, cap=256: nat.lib.hg_cram_crai_slice(C.cast(C.c_char_p(multi), _vp), len(multi), 3, r.ctypes.data, a.ctypes.data, e.ctypes.data, 1, 2, 3, buf, cap)
    n = call(rid, ap, ae)
    assert buf.raw[:n] == b"0\t10\t50\t1\t2\t3\n1\t5\t100\t1\t2\t3\n-1\t0\t1\t1\t2\t3\n"          # the span reaches the furthest end of the run, not the last one
    assert call(rid, np.array([10, 9, 5, 0], np.int64), ae) == -2
    assert call(rid, ap, ae, cap=20) < 0                                   # the text does not fit


@pytest.mark.gpu
def test_gpu_whole_cram_file_to_bam(engine):
    """hg_cram_file_to_bam_host: the reference's 34 CRAM files, byte for byte as they sit in its test directory, come back as an
    uncompressed BAM stream -- header with the file's @SQ lines, then the records of the twin (container walk, block CRCs and
    decompression, record decoding, cram_to_bam: all inside the one call)."""
    import struct
    from htslib_amd import _native as nat

    class RefSeq(C.Structure):
        _fields_ = [("bases", _vp), ("len", C.c_uint64)]

    nrec = md5_refused = 0
    for f in json.load(open(GOLD)):
        cram = unpack(f["cram"])
        spans = {}
        for s in f["slices"]:
            for t, a, b, ln in s["refs"]:
                spans.setdefault(t, (ln, []))[1].append((a, unpack(b)))
        seqs = []
        for i, name in enumerate(f["ref_names"]):
            if f["full_refs"]: seqs.append(bytearray(unpack(dict(f["full_refs"])[name])))
            elif i in spans:                                                 # paste the stored stretches into an all-N sequence of the @SQ length
                sq = bytearray(b"N" * spans[i][0])
                for a, b in spans[i][1]: sq[a - 1:a - 1 + len(b)] = b
                seqs.append(sq)
            else: seqs.append(None)
        keep = [C.create_string_buffer(bytes(q), len(q)) if q is not None else None for q in seqs]
        arr = (RefSeq * max(len(seqs), 1))(*[RefSeq(C.addressof(k), len(q)) if k is not None else RefSeq(None, 0) for k, q in zip(keep, seqs)])
        out = np.zeros(1 << 22, np.uint8); total = C.c_uint64(); n = C.c_uint64()
        cb = C.create_string_buffer(cram, len(cram))
        rc = nat.lib.hg_cram_file_to_bam_host(engine._h, C.cast(cb, _vp), len(cram), C.cast(arr, _vp), len(seqs), out.ctypes.data, len(out), C.byref(total), C.byref(n))
        assert rc == 0, (f["file"], rc)
        b = bytes(out[:total.value])
        assert b[:4] == b"BAM\x01"
        lt = struct.unpack_from("<i", b, 4)[0]; p = 8 + lt
        nref = struct.unpack_from("<i", b, p)[0]; p += 4
        names = []
        for _ in range(nref):
            ln = struct.unpack_from("<i", b, p)[0]; names.append(b[p + 4:p + 4 + ln - 1].decode()); p += 4 + ln + 4
        assert names == f["ref_names"], f["file"]
        recs = _parse_bam_records(b[p:])
        expect = [e for s in f["slices"] for e in s["expect"]]
        assert len(recs) == len(expect) == n.value, f["file"]
        for (g, bn, raw), e in zip(recs, expect):
            check_against_twin(f["file"], [g], [e]); nrec += 1
        # the wrong reference: one base changed inside every stored stretch -> "MD5 checksum reference mismatch" (cram_decode.c:2480-2540) for
        # files whose slice headers carry a digest; the reference's ignore_md5 option decodes them all the same
        wrong = [bytearray(q) if q is not None else None for q in seqs]
        for i in spans:
            for a, bb in spans[i][1]:
                if i < len(wrong) and wrong[i] is not None and len(bb): wrong[i][a - 1 + len(bb) // 2] ^= 0x06      # A<->G, C<->E ...: never the same base
        keepw = [C.create_string_buffer(bytes(q), len(q)) if q is not None else None for q in wrong]
        arrw = (RefSeq * max(len(wrong), 1))(*[RefSeq(C.addressof(k), len(q)) if k is not None else RefSeq(None, 0) for k, q in zip(keepw, wrong)])
        rcw = nat.lib.hg_cram_file_to_bam_host(engine._h, C.cast(cb, _vp), len(cram), C.cast(arrw, _vp), len(wrong), out.ctypes.data, len(out), C.byref(total), C.byref(n))
        rci = nat.lib.hg_cram_file_to_bam_host2(engine._h, C.cast(cb, _vp), len(cram), C.cast(arrw, _vp), len(wrong), out.ctypes.data, len(out), C.byref(total), C.byref(n), 1, None)
        assert rci == 0, (f["file"], rci)
        md5_refused += int(rcw != 0)
    assert nrec == 230
    assert md5_refused >= 3, md5_refused
    # a damaged file: one payload byte of the last data block flipped -> the block's CRC fails the file
    bad = bytearray(cram); bad[len(bad) // 2] ^= 0x10
    cb = C.create_string_buffer(bytes(bad), len(bad))
    assert nat.lib.hg_cram_file_to_bam_host(engine._h, C.cast(cb, _vp), len(bad), C.cast(arr, _vp), len(seqs), out.ctypes.data, len(out), C.byref(total), C.byref(n)) != 0


def test_data_parallel_prototype_matches_the_chain_decoder(hostlib, tmp_path):
    """tests/native/cram_fastpath_proto.cpp: the scan-and-map formulation planned for the device -- whole-block column decodes,
    prefix sums for every "which item does this record / feature read" question, one independent walk per record -- gives exactly the chain
    decoder's columns, CIGARs, names, bases and qualities on production-size EXTERNAL-only slices."""
    from htslib_amd import synth_cram
    so = str(tmp_path / "libproto.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-o", so, os.path.join(ROOT, "tests", "native", "cram_fastpath_proto.cpp")], check=True)
    Pr = C.CDLL(so)
    Pr.hgr_proto_decode_slice.argtypes = [_vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp]
    rng = np.random.default_rng(31)
    slices = [synth_cram.make_slice(rng, 3000, 100), synth_cram.make_slice(rng, 500, 151, unmapped_every=3, detached_every=4), synth_cram.make_slice(rng, 1, 40, ref_len=500),
              synth_cram.make_slice(rng, 257, 75, unmapped_every=0, detached_every=0), synth_cram.make_slice(rng, 1200, 100, tags=True),
              synth_cram.make_slice(rng, 33, 60, unmapped_every=2, tags=True)]
    DECODE_MD[0] = 0                                                  # the prototype does not regenerate MD / NM: compare the stored tags
    try:
        st, chain = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 1)
    finally:
        DECODE_MD[0] = -1
    assert (st == 0).all()
    for s, want in zip(slices, chain):
        keep = []
        arr = _slice_array([s], keep)
        R = s["nrec"]
        i32 = {k: np.full(R, -99, np.int32) for k in ("flags", "cram_flags", "ref_id", "len", "rg", "mqual", "mate_ref_id", "ncigar", "name_len")}
        i64 = {k: np.full(R, -99, np.int64) for k in ("apos", "aend", "mate_pos", "tlen")}
        u64 = {k: np.zeros(R, np.uint64) for k in ("cigar_off", "name_off")}
        cigar = np.zeros(R * 16 + 64, np.uint32); names = np.zeros(R * 16 + 64, np.uint8)
        ncap = sum(len(e[9]) for e in s["expect"]) + 64
        seq_off = np.zeros(R, np.uint64); seq = np.zeros(ncap, np.uint8); qual = np.zeros(ncap, np.uint8)
        aux_off = np.zeros(R, np.uint64); aux_len = np.zeros(R, np.int32); aux = np.zeros(R * 64 + 64, np.uint8)
        cols = Cols(*[a.ctypes.data for a in list(i32.values()) + list(i64.values()) + list(u64.values()) + [cigar, names]], seq_off.ctypes.data, seq.ctypes.data, qual.ctypes.data,
                    aux_off.ctypes.data, aux_len.ctypes.data, aux.ctypes.data)
        used = np.zeros(4, np.uint64)
        rc = Pr.hgr_proto_decode_slice(C.cast(arr, _vp), 3, 1, len(cigar), len(names), ncap, len(aux), C.byref(cols), used.ctypes.data)
        assert rc == 0, rc
        for r in range(R):
            co, nc = int(u64["cigar_off"][r]), int(i32["ncigar"][r]); no, nl = int(u64["name_off"][r]), int(i32["name_len"][r]); so_, ln = int(seq_off[r]), int(i32["len"][r])
            q = qual[so_:so_ + ln]
            got = [bytes(names[no:no + nl]).decode(), int(i32["flags"][r]), int(i32["ref_id"][r]), int(i64["apos"][r]), int(i32["mqual"][r]),
                   [[int(c >> 4), int(c & 15)] for c in cigar[co:co + nc]], int(i32["mate_ref_id"][r]), int(i64["mate_pos"][r]), int(i64["tlen"][r]),
                   bytes(seq[so_:so_ + ln]).decode("latin1") if ln else "*", "*" if ln == 0 or (q == 255).all() else bytes((q + 33).astype(np.uint8)).decode("latin1")]
            got.append([G.short_tag(t) for t in G.aux_to_text(bytes(aux[int(aux_off[r]):int(aux_off[r]) + int(aux_len[r])]))])
            assert got == want[r][:12], (r, got, want[r][:12])
    # a slice the scheme does not cover (CORE-coded series, shared blocks: the reference's fixtures) is handed back
    fx = [x for _, _, _, x in load_slices()][0]
    keep = []
    assert Pr.hgr_proto_decode_slice(C.cast(_slice_array([fx], keep), _vp), 3, 1, 64, 64, 64, 64, C.byref(cols), used.ctypes.data) == -3


def test_damaged_inputs_are_rejected_or_decoded_never_fatal(hostlib):
    """400 mutated slices (compression header, slice header, CORE, EXTERNAL blocks: flipped bits, truncations, overwritten and inserted
    bytes; halved reference spans) through the decoder source: each comes back with a status -- the same walk ran 10 000 times under
    AddressSanitizer while the decoder was written; on the device an out-of-bounds read would be a fault, not an error code."""
    from htslib_amd import synth_cram
    rng = np.random.default_rng(77)
    base = [s for _, _, _, s in load_slices()] + [synth_cram.make_slice(rng, 40, 60, tags=True)]

    def mutate(b):
        b = bytearray(b)
        if not b: return bytes(b)
        k = int(rng.integers(0, 4))
        if k == 0:
            for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1: b = b[:int(rng.integers(0, len(b)))]
        elif k == 2: b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:
            i = int(rng.integers(0, len(b))); b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8))
        return bytes(b)

    ok = bad = 0
    for it in range(400):
        s = dict(base[int(rng.integers(0, len(base)))])
        what = int(rng.integers(0, 5))
        if what == 0: s["comp_hdr"] = mutate(s["comp_hdr"])
        elif what == 1: s["slice_hdr"] = mutate(s["slice_hdr"][:3]) + s["slice_hdr"][3:] if it % 2 else mutate(s["slice_hdr"])
        elif what == 2: s["core"] = mutate(s["core"])
        elif what == 3 and s["blocks"]:
            j = int(rng.integers(0, len(s["blocks"]))); bl = list(s["blocks"]); bl[j] = (bl[j][0], mutate(bl[j][1])); s["blocks"] = bl
        else: s["refs"] = [(t, a, b[:len(b) // 2], ln) for t, a, b, ln in s.get("refs", [])]
        try:
            st, got = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, [s], 3, 7)
            ok += int(st[0] == 0); bad += int(st[0] != 0)
        except (AssertionError, ValueError, IndexError, UnicodeDecodeError, struct_error):      # the harness refusing the batch, or tag bytes that are not BAM aux (damaged, yet "decoded")
            bad += 1
    assert ok > 50 and bad > 50


def test_encode_prototype_round_trips_through_the_pinned_decoder(hostlib, tmp_path):
    """tests/native/cram_encode_proto.cpp: the ENCODE side in column form (records -> features -> one EXTERNAL column per series + headers).
    Every slice of the reference's fixtures and tagged synthetic slices is decoded, re-encoded, decoded again: names, flags, positions,
    CIGARs, mates, template lengths, bases, qualities and tags all survive."""
    import struct
    from htslib_amd import synth_cram
    so = str(tmp_path / "libenc.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-o", so, os.path.join(ROOT, "tests", "native", "cram_encode_proto.cpp")], check=True)
    E = C.CDLL(so)
    E.hgr_proto_reencode_slice.restype = C.c_long
    E.hgr_proto_reencode_slice.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_size_t]
    rng = np.random.default_rng(41)
    cases = [(fname, nref, s) for fname, major, nref, s in load_slices()]
    cases += [("synthetic", 1, synth_cram.make_slice(rng, 1500, 100, tags=True)), ("synthetic", 1, synth_cram.make_slice(rng, 60, 151, unmapped_every=3))]
    DECODE_MD[0] = 0
    try:
        done = 0; skipped = []
        for fname, nref, s in cases:
            keep = []
            arr = _slice_array([s], keep)
            buf = np.zeros(1 << 22, np.uint8)
            n = E.hgr_proto_reencode_slice(C.cast(arr, _vp), 3, nref, buf.ctypes.data, len(buf))
            if n == -3: skipped.append(fname); continue                  # a CIGAR without bases (SEQ "*"): outside the prototype
            assert n > 0, (fname, n)
            b = bytes(buf[:n]); p = 0
            def take():
                nonlocal p
                ln = struct.unpack_from("<I", b, p)[0]; p += 4; v = b[p:p + ln]; p += ln; return v
            comp, sh = take(), take()
            nb = struct.unpack_from("<I", b, p)[0]; p += 4
            blocks = []
            for _ in range(nb):
                cid, ln = struct.unpack_from("<iI", b, p); p += 8; blocks.append((cid, b[p:p + ln])); p += ln
            s2 = {"comp_hdr": comp, "slice_hdr": sh, "core": b"", "blocks": blocks, "nrec": s["nrec"], "refs": s["refs"], "expect": s["expect"]}
            st1, one = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, [s], 3, nref)
            st2, two = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, [s2], 3, nref)
            assert st1[0] == 0 and st2[0] == 0, (fname, st1, st2)
            assert one[0] == two[0], (fname, [(x, y) for x, y in zip(one[0], two[0]) if x != y][:2])
            done += len(one[0])
        assert done >= 200 + 1560 and len(skipped) <= 6, (done, skipped)
    finally:
        DECODE_MD[0] = -1


def test_columns_only_mode_without_bases_qualities_and_tags(hostlib):
    """The caller may ask for the record columns alone (seq / qual / aux pointers NULL, no reference spans): same fields, nothing else touched."""
    slices = [s for fname, major, nref, s in load_slices() if fname == "test/range.cram"]
    st, full = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 7)
    st2, cols = decode(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 7, with_seq=False)
    assert (st == 0).all() and (st2 == 0).all()
    for a, b in zip(full, cols):
        assert [r[:9] for r in a] == [r[:9] for r in b] and all(len(r) == 9 for r in b)


@pytest.mark.gpu
def test_gpu_cram_index_build_whole_files(engine):
    """hg_cram_index_build_host = cram_index_build (cram/cram_index.c:779-870): the file walk + one line per slice; test/range.cram (three multi-reference
    slices: their records are decoded for the per-reference runs) gives the reference's own range.cram.crai byte for byte; every other fixture gives the
    lines its slice headers dictate at the container / slice offsets recorded when the fixtures were frozen."""
    from htslib_amd import _native as nat
    checked = 0
    for f in json.load(open(GOLD)):
        cram = unpack(f["cram"])
        cb = C.create_string_buffer(cram, len(cram)); out = C.create_string_buffer(1 << 16)
        n = nat.lib.hg_cram_index_build_host(engine._h, C.cast(cb, _vp), len(cram), out, 1 << 16)
        assert n >= 0, (f["file"], n)
        text = out.raw[:n].decode()
        if f["crai"]:
            assert text == f["crai"], f["file"]; checked += 1
            continue
        want = ""
        for s in f["slices"]:
            sh = unpack(s["slice_hdr"])
            buf = C.create_string_buffer(4096)
            k = nat.lib.hg_cram_crai_slice(C.cast(C.c_char_p(sh), _vp), len(sh), f["major"], None, None, None, s["cpos"], s["landmark"], s["slice_bytes"], buf, 4096)
            if k < 0: want = None; break                                 # a multi-reference slice: only range.cram has a reference-written index to compare with
            want += buf.raw[:k].decode()
        if want is not None:
            assert text == want, f["file"]; checked += 1
    assert checked >= 30
