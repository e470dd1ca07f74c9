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


def _check_we_write_they_read(engine, tmp_path, bam, names, seqs, per_slice, tag, md_default=False):
    rc, cram, n = _bam_to_cram(engine, bam, seqs, per_slice)
    assert rc == 0, (tag, rc)
    p = str(tmp_path / (tag + ".cram")); open(p, "wb").write(cram)
    fa = RC.write_fasta(str(tmp_path / (tag + ".fa")), names, seqs) if any(s is not None for s in seqs) else None
    src = RC.write_bam_file(str(tmp_path / (tag + ".bam")), bam)
    hw, want = RC.sam_records(src)
    hg, got = RC.sam_records(p, fa, extra=() if md_default else ("-i", "decode_md=0"))
    assert len(want) == n
    # the RG:Z tag of a record becomes the RG series and comes back at the end of the tag list (cram_decode.c:3178-3190): compare with RG moved there
    def norm(l):
        f = l.split(b"\t")
        rg = [t for t in f[11:] if t.startswith(b"RG:Z:")]
        return f[:11] + [t for t in f[11:] if not t.startswith(b"RG:Z:")] + rg
    d = RC.first_difference([norm(l) for l in got], [norm(l) for l in want])
    assert d is None, (tag, d)
    assert [l for l in hg if l[:3] in (b"@SQ", b"@RG")] and len([l for l in hg if l.startswith(b"@SQ")]) == len(names)
    return len(cram)


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


def _check_they_write_we_read(engine, tmp_path, src_bam, fa, names, seqs, opts, tag, threads=4):
    p = RC.to_cram(src_bam, str(tmp_path / (tag + ".cram")), fa, opts=("version=3.0",) + tuple(opts), threads=threads)
    ht, theirs = RC.sam_records(p, fa)
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
