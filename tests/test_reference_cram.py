"""CRAM files against STOCK htslib -- the reference's own container / slice / record reader and writer (cram_decode_slice, cram_encode_slice,
cram_compress_block3 ... compiled from /root/reference by oracle/Makefile into oracle/_ref/libref_hts.so, driven through the reference's own
test/test_view.c = oracle/_ref/ref_view; the absent htscodecs submodule is replaced by oracle/htscodecs_stub, so this is a CRAM <= 3.0 tool whose
only non-reference code is the PINNED rANS 4x8 restatement).

  (a) north_star's bar for a writer -- "htslib-decodable output": what hg_bam_to_cram_host writes (record encoder on the device, block auto-tuner
      with GZIP / rANS 4x8, container framing, RR / slice MD5) is read by ref_view and comes out as the records that went in;
  (b) what the reference WRITES at production slice size is decoded by hg_cram_file_to_bam_host2 (data-parallel passes + cram_to_bam) to exactly what the
      reference's own cram_decode_slice makes of it -- SAM text of both, MD / NM regeneration included."""
import os
import struct

import numpy as np
import pytest

from tests import refcram as RC
from tests.test_cram_encode import _bam_to_cram, _file_to_bam

needs_ref_view = pytest.mark.skipif(not RC.have(), reason="oracle/_ref/ref_view not built (make -C oracle ref needs /root/reference)")


def _synthetic(engine, nslices, nrec, tags=True, seed=11, readlen=150):
    """slices of htslib_amd/synth_cram.py (reads aligned to a random reference of their own: clips, substitutions, insertions, deletions, unmapped and detached
    records, tags) -> the pinned record decoder -> BAM records; slice k's records sit on reference k -> (BAM stream with header, names, sequences)"""
    from htslib_amd import synth_cram
    rng = np.random.default_rng(seed)
    sl = [synth_cram.make_slice(rng, nrec, readlen, tags=tags) for _ in range(nslices)]
    bam, names, seqs, _ = synth_cram.bam_from_slices(engine, sl)
    return bam, names, seqs


@needs_ref_view
def test_ref_view_round_trips_through_its_own_cram30_writer(tmp_path):
    """sanity of the checker itself, no GPU: BAM -> (reference writer, rANS 4x8 from the pinned restatement) CRAM 3.0 -> SAM == BAM -> SAM"""
    from htslib_amd import synth
    plain, _, _ = synth.bam_stream(3 << 20, 0x5EED0001, 0, True)
    q = RC.header_len(plain)
    while q + 4 <= len(plain) and q + 4 + struct.unpack_from("<i", plain, q)[0] <= len(plain): q += 4 + struct.unpack_from("<i", plain, q)[0]
    bam = RC.write_bam_file(str(tmp_path / "in.bam"), plain[:q])
    RC.to_cram(bam, str(tmp_path / "out.cram"), opts=("version=3.0", "no_ref=1", "seqs_per_slice=2000"))
    h1, r1 = RC.sam_records(bam)
    h2, r2 = RC.sam_records(str(tmp_path / "out.cram"))
    assert len(r1) > 5000 and r1 == r2
    assert open(tmp_path / "out.cram", "rb").read(6) == b"CRAM\x03\x00"
    # the stand-in offers no CRAM 3.1 codec: a 3.1 writer's Nx16 / tok3 trials fail and cram_compress_block3 falls back to gzip (its "redoing trial" path)
    rc, _, err = RC.run(["-C", "-o", "version=3.1", "-o", "no_ref=1", "-p", str(tmp_path / "v31.cram"), bam])
    assert rc == 0 and "failed, redoing trial" in err and open(tmp_path / "v31.cram", "rb").read(6) == b"CRAM\x03\x01"
    assert RC.sam_records(str(tmp_path / "v31.cram"))[1] == r1


def _check_we_write_they_read(engine, tmp_path, bam, names, seqs, per_slice, tag, md_default=False, flags=0, level=5, env=None):
    rc, cram, n = _bam_to_cram(engine, bam, seqs, per_slice, flags, level)
    assert rc == 0, (tag, rc)
    assert cram[:6] == b"CRAM\x03" + bytes([1 if flags & 1 else 0])
    p = str(tmp_path / (tag + ".cram")); open(p, "wb").write(cram)
    fa = RC.write_fasta(str(tmp_path / (tag + ".fa")), names, seqs) if any(s is not None for s in seqs) else None
    src = RC.write_bam_file(str(tmp_path / (tag + ".bam")), bam)
    hw, want = RC.sam_records(src)
    hg, got = RC.sam_records(p, fa, extra=() if md_default else ("-i", "decode_md=0"), env=env)
    assert len(want) == n
    # the RG:Z tag of a record becomes the RG series and comes back at the end of the tag list (cram_decode.c:3178-3190): compare with RG moved there
    def norm(l):
        f = l.split(b"\t")
        rg = [t for t in f[11:] if t.startswith(b"RG:Z:")]
        return f[:11] + [t for t in f[11:] if not t.startswith(b"RG:Z:")] + rg
    d = RC.first_difference([norm(l) for l in got], [norm(l) for l in want])
    assert d is None, (tag, d)
    assert [l for l in hg if l[:3] in (b"@SQ", b"@RG")] and len([l for l in hg if l.startswith(b"@SQ")]) == len(names)
    return cram


@pytest.mark.gpu
@needs_ref_view
def test_the_reference_reader_reads_the_cram31_files_we_write(engine, tmp_path):
    """BASELINE configs[4], "full CRAM 3.1 encode": hg_bam_to_cram_host2 with HG_CRAM_WRITE_V31 (+ HG_CRAM_WRITE_ARITH) -- record encoder on the device, every series
    through the block auto-tuner with the method sets cram_compress_slice gives a 3.1 writer (rANS Nx16 variants, tok3 for the names, the range coder, gzip).  The
    reference's READER (ref_view with the htscodecs stand-in on oracle/'s codecs: the codec dialect is unpinned, the container / block / record layers are the
    reference's) must give back the records that went in; so must our own decoder; and the 3.1 methods must actually have been chosen."""
    bam, names, seqs = _synthetic(engine, 4, 10000, seed=13)
    E = {"ORC_STUB_CODECS31": "1"}
    seen = {}
    for tag, flags, level, per in (("v31", 1, 5, 10000), ("v31_arith", 3, 5, 10000), ("v31_l7_multi", 3, 7, 7000), ("v31_l1", 1, 1, 10000)):
        cram = _check_we_write_they_read(engine, tmp_path, bam, names, seqs, per, tag, flags=flags, level=level, env=E)
        for m, c in _block_methods(cram).items(): seen[m] = seen.get(m, 0) + c
        rc, back, n = _file_to_bam(engine, cram, seqs)
        assert rc == 0 and n == 40000, (tag, rc)
    assert seen.get(5, 0) > 20 and seen.get(8, 0) >= 4 and seen.get(6, 0) > 0 and seen.get(4, 0) == 0, seen      # rANS Nx16, tok3, range coder; no 4x8 in a 3.1 file


@pytest.mark.gpu
@needs_ref_view
def test_stock_htslib_reads_the_cram_files_we_write(engine, tmp_path):
    # 1. production-size slices on references: one slice per reference (single-reference slices with a digest), then cut so that slices span references
    bam, names, seqs = _synthetic(engine, 4, 10000)
    _check_we_write_they_read(engine, tmp_path, bam, names, seqs, 10000, "single_ref")
    _check_we_write_they_read(engine, tmp_path, bam, names, seqs, 7000, "multi_ref")
    _check_we_write_they_read(engine, tmp_path, bam, names, seqs, 10000, "md_default", md_default=True)
    # 2. the same records with NO reference handed over (RR = 0: every base stored), and with only some of them
    _check_we_write_they_read(engine, tmp_path, bam, names, [None] * 4, 10000, "no_ref")
    _check_we_write_they_read(engine, tmp_path, bam, names, [seqs[0], None, seqs[2], None], 6000, "some_refs")
    # 3. the bench's BAM generator: 25 @SQ, pairs, NM / MD / RG tags, no reference
    from htslib_amd import synth
    plain, _, _ = synth.bam_stream(6 << 20, 0x5EED0001, 0, True)
    q = RC.header_len(plain)
    while q + 4 <= len(plain) and q + 4 + struct.unpack_from("<i", plain, q)[0] <= len(plain): q += 4 + struct.unpack_from("<i", plain, q)[0]
    refs = RC.header_refs(plain)
    _check_we_write_they_read(engine, tmp_path, plain[:q], [r[0] for r in refs], [None] * len(refs), 3000, "bench_bam")


@pytest.mark.gpu
@needs_ref_view
def test_stock_htslib_reads_our_rewrite_of_the_reference_fixtures(engine, tmp_path):
    """the reference's CRAM fixtures -> BAM stream (pinned decoder) -> our CRAM 3.0 writer -> ref_view == ref_view of the BAM stream"""
    import json
    from tests import test_cram_records as T
    done = 0
    for f in json.load(open(T.GOLD)):
        cram = T.unpack(f["cram"])
        spans = {}
        for s in f["slices"]:
            for t, a, b, ln in s["refs"]:
                spans.setdefault(t, (ln, []))[1].append((a, T.unpack(b)))
        seqs = []
        for i, name in enumerate(f["ref_names"]):
            if f["full_refs"]: seqs.append(bytes(T.unpack(dict(f["full_refs"])[name])))
            elif i in spans:
                sq = bytearray(b"N" * spans[i][0])
                for a, b in spans[i][1]: sq[a - 1:a - 1 + len(b)] = b
                seqs.append(bytes(sq))
            else: seqs.append(None)
        rc, bam, n = _file_to_bam(engine, cram, seqs)
        assert rc == 0, f["file"]
        rc2, _, _ = _bam_to_cram(engine, bam, seqs)
        if rc2 == -6: continue                                            # a slice the encoder does not cover
        tag = os.path.basename(f["file"]).replace("#", "_")
        _check_we_write_they_read(engine, tmp_path, bam, list(f["ref_names"]), seqs, 0, tag)
        done += 1
    assert done >= 28, done


def _check_they_write_we_read(engine, tmp_path, src_bam, fa, names, seqs, opts, tag, threads=4, version="3.0", env=None):
    p = RC.to_cram(src_bam, str(tmp_path / (tag + ".cram")), fa, opts=("version=" + version,) + tuple(opts), threads=threads, env=env)
    assert open(p, "rb").read(6) == b"CRAM" + bytes([int(version[0]), int(version[2])])
    ht, theirs = RC.sam_records(p, fa, env=env)
    rc, ours, n = _file_to_bam(engine, open(p, "rb").read(), seqs)
    assert rc == 0 and n == len(theirs), (tag, rc, n, len(theirs))
    ho, got = RC.sam_records(RC.write_bam_file(str(tmp_path / (tag + ".ours.bam")), ours))
    d = RC.first_difference(got, theirs)
    assert d is None, (tag, d)
    assert ho == ht, (tag, RC.first_difference(ho, ht))
    return n


@pytest.mark.gpu
@needs_ref_view
def test_we_read_the_cram_files_stock_htslib_writes(engine, tmp_path):
    bam, names, seqs = _synthetic(engine, 8, 10000, seed=5)
    fa = RC.write_fasta(str(tmp_path / "ref.fa"), names, seqs)
    src = RC.write_bam_file(str(tmp_path / "in.bam"), bam)
    n = _check_they_write_we_read(engine, tmp_path, src, fa, names, seqs, (), "default")                       # 10 000 records per slice, one reference per slice
    assert n == 80000
    _check_they_write_we_read(engine, tmp_path, src, fa, names, seqs, ("seqs_per_slice=3000",), "s3000")
    _check_they_write_we_read(engine, tmp_path, src, fa, names, seqs, ("seqs_per_slice=3000", "multi_seq_per_slice=1"), "multi")
    _check_they_write_we_read(engine, tmp_path, src, fa, names, seqs, ("seqs_per_slice=4000", "slices_per_container=3"), "spc3")
    _check_they_write_we_read(engine, tmp_path, src, None, names, [None] * len(seqs), ("no_ref=1",), "no_ref")
    _check_they_write_we_read(engine, tmp_path, src, fa, names, seqs, ("embed_ref=1",), "embed_ref")
    _check_they_write_we_read(engine, tmp_path, src, fa, names, seqs, ("level=1", "use_bzip2=0"), "level1")
    _check_they_write_we_read(engine, tmp_path, src, fa, names, seqs, ("level=9",), "level9")


def _block_methods(cram: bytes):
    """the on-disk method byte of every block of a CRAM 3.x file (container walk: cram_read_container / cram_read_block, cram_io.c:3590-3760, 1414-1483)"""
    def itf8(b, p):
        c = b[p]
        if c < 0x80: return c, p + 1
        if c < 0xc0: return ((c & 0x3f) << 8) | b[p + 1], p + 2
        if c < 0xe0: return ((c & 0x1f) << 16) | (b[p + 1] << 8) | b[p + 2], p + 3
        if c < 0xf0: return ((c & 0x0f) << 24) | (b[p + 1] << 16) | (b[p + 2] << 8) | b[p + 3], p + 4
        return ((c & 0x0f) << 28) | (b[p + 1] << 20) | (b[p + 2] << 12) | (b[p + 3] << 4) | (b[p + 4] & 0x0f), p + 5
    def ltf8(b, p):
        c = b[p]; n = 0
        while n < 8 and c & (0x80 >> n): n += 1
        return 0, p + 1 + n
    methods = {}
    p = 26
    while p < len(cram):
        length = struct.unpack_from("<i", cram, p)[0]; q = p + 4
        for _ in range(3): _, q = itf8(cram, q)                           # ref id, start, span
        _, q = itf8(cram, q)                                             # records
        _, q = ltf8(cram, q); _, q = ltf8(cram, q)                       # record counter, bases
        nblk, q = itf8(cram, q); nland, q = itf8(cram, q)
        for _ in range(nland): _, q = itf8(cram, q)
        q += 4                                                           # header CRC
        end = q + length
        while q < end:
            m = cram[q]; q += 2
            _, q = itf8(cram, q); csz, q = itf8(cram, q); _, q = itf8(cram, q)
            methods[m] = methods.get(m, 0) + 1
            q += csz + 4
        p = end
    return methods


@pytest.mark.gpu
@needs_ref_view
def test_we_read_cram31_files_of_the_reference_writer_with_the_unpinned_codecs(engine, tmp_path):
    """CRAM 3.1 END TO END, as far as this box allows.  The reference's writer (cram_encode_slice, cram_compress_block3, the fqz_slice it builds, its tok3 levels and
    flag maps: cram_io.c:1756-1904) runs with the htscodecs stand-in switched to oracle/'s restatements of methods 5-8 (ORC_STUB_CODECS31=1) -- the codec DIALECT
    stays unpinned (htscodecs is absent), but everything around it is the reference's: which method and flags each series gets, block framing, what the
    quality codec is told about the records.  Our whole-file decoder (device codecs for rANS Nx16 / range coder / fqzcomp / tok3 + the record passes) must
    give exactly what the reference's own reader makes of the same file, for every profile the reference offers."""
    bam, names, seqs = _synthetic(engine, 4, 10000, seed=9)
    # qualities the way a sequencer writes them (a slow walk that decays along the read): with the generator's uniform noise no quality codec ever wins a trial
    b = bytearray(bam); at = RC.header_len(bam); rng = np.random.default_rng(3)
    while at < len(b):
        ln = struct.unpack_from("<i", b, at)[0]
        l_name, n_cig, l_seq = b[at + 12], struct.unpack_from("<H", b, at + 16)[0], struct.unpack_from("<i", b, at + 20)[0]
        q0 = at + 36 + l_name + 4 * n_cig + (l_seq + 1) // 2
        walk = np.cumsum(rng.integers(-1, 2, l_seq)) - np.arange(l_seq) * (12.0 / max(l_seq, 1))
        b[q0:q0 + l_seq] = np.clip(38 + walk, 2, 40).astype(np.uint8).tobytes()
        at += 4 + ln
    bam = bytes(b)
    fa = RC.write_fasta(str(tmp_path / "ref.fa"), names, seqs)
    src = RC.write_bam_file(str(tmp_path / "in.bam"), bam)
    E = {"ORC_STUB_CODECS31": "1"}
    seen = {}
    for tag, opts in (("normal", ()), ("fast", ("fast",)), ("small", ("small",)), ("archive", ("archive",)),
                      ("arith_fqz_tok", ("use_arith=1", "use_fqz=1", "use_tok=1", "level=7")), ("s3000_multi", ("seqs_per_slice=3000", "multi_seq_per_slice=1", "use_arith=1"))):
        _check_they_write_we_read(engine, tmp_path, src, fa, names, seqs, opts, "v31_" + tag, version="3.1", env=E)
        for m, c in _block_methods(open(tmp_path / ("v31_" + tag + ".cram"), "rb").read()).items(): seen[m] = seen.get(m, 0) + c
    assert seen.get(5, 0) > 20 and seen.get(8, 0) > 0 and seen.get(6, 0) > 0 and seen.get(7, 0) > 0, seen     # rANS Nx16, tok3, range coder, fqzcomp blocks were all there
