"""THE DROP-IN INSIDE libhts: the reference's own programs on oracle/_ref/libhts_gpu.so.

libhts_gpu.so (oracle/Makefile, target libhts_gpu) = every object of the reference's libhts except bgzf.o, with cram_io.o's / cram_external.o's
block functions weakened, plus OUR bgzf_front.cpp, cram_block_front.cpp and htscodecs_front.cpp on libhtsgpu.so.  No oracle, no htscodecs stand-in, no
bgzf.c.  The reference's test/test_view.c (a small `samtools view`), test/test_index.c and test/test_bgzf.c are linked to it UNMODIFIED, so here

  * sam.c's bam_read1 / bam_write1 and the on-the-fly index (sam.c:784-928,942-943 -> bgzf_idx_push, hts.c:2558-2640) run on our bgzf_read / bgzf_write,
  * cram_decode_slice's block loop (cram/cram_decode.c:624-627) runs on our cram_uncompress_block,
  * cram_encode_slice / cram_compress_slice (cram/cram_encode.c:803-988) run on our cram_compress_block2 + cram_metrics,
  * cram_read_container / cram_flush_container run on our cram_read_block / cram_write_block,

driven the way the reference's harness drives them: test/test.pl:708-830 (test_view: SAM -> BAM / CRAM 2.1 / 3.0 / 3.1 -> SAM over every test/*#*.sam, the
pre-made htsjdk CRAMs) and test/test.pl:1067-1160 (test_index: indexes written on the fly and by test_index against the reference's golden .bai / .csi /
.crai / .tbi).  The checker is STOCK htslib: oracle/_ref/ref_view (the same program on the reference's whole libhts) must print the same SAM text for every
file either side writes, in every direction.  Fixtures: tests/golden/view_fixtures.tar.gz (the reference's test data, made by tests/golden/make_view_fixtures.sh)."""
import gzip
import hashlib
import os
import subprocess
import tarfile

import pytest

from tests import dropin_cases, refutil

REF = refutil.REF_DIR
VIEW_GPU = os.path.join(REF, "ref_view_gpu")
VIEW_REF = os.path.join(REF, "ref_view")
INDEX_GPU = os.path.join(REF, "test_index_gpu")
BGZF_GPU = os.path.join(REF, "test_bgzf_libhts_gpu")
FIX = os.path.join(refutil.ROOT, "tests", "golden", "view_fixtures.tar.gz")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(VIEW_GPU) and os.path.exists(VIEW_REF) and os.path.exists(os.path.join(REF, "libhts_gpu.so"))),
                                 reason="oracle/_ref/libhts_gpu.so + ref_view_gpu + ref_view not built (make -C oracle ref needs /root/reference)")]

# test/test.pl:712-736
CRAM31 = {"auxf#values.sam", "c1#pad3.sam", "ce#5.sam", "ce#1000.sam", "ce#large_seq.sam", "ce#supp.sam", "xx#MD.sam", "xx#blank.sam", "xx#large_aux.sam",
          "xx#pair.sam", "xx#tlen.sam"}
CRAM_MS = {"ce#1000.sam", "ce#5.sam", "ce#5b.sam", "ce#unmap.sam", "ce#unmap1.sam", "ce#unmap2.sam", "xx#blank.sam", "xx#minimal.sam", "xx#tlen.sam",
           "xx#tlen2.sam", "xx#triplet.sam"}


def unpack_fixtures(d):
    with tarfile.open(FIX) as t:
        t.extractall(d)
    # the MD5 reference cache test.pl builds with ce_fa_to_md5_cache (REF_PATH=<dir>/%2s/%2s/%s)
    m5 = os.path.join(d, "md5")
    for fa in ["ce.fa"]:
        for rec in open(os.path.join(d, fa)).read().split(">")[1:]:
            seq = "".join(rec.split("\n")[1:]).upper().encode()
            h = hashlib.md5(seq).hexdigest()
            os.makedirs(os.path.join(m5, h[:2], h[2:4]), exist_ok=True)
            open(os.path.join(m5, h[:2], h[2:4], h[4:]), "wb").write(seq)
    return d


@pytest.fixture(scope="module")
def fx(tmp_path_factory):
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU")
    return unpack_fixtures(str(tmp_path_factory.mktemp("viewfix")))


def _env(d, md5_cache=False):
    """test.pl points REF_PATH at the MD5 cache of ce.fa for test_index only (:1097); everywhere else references come from -t / the header's UR"""
    e = dict(os.environ, REF_CACHE="", HTS_GPU_STRICT="1")
    e.pop("ORC_STUB_CODECS31", None); e.pop("REF_PATH", None)
    if md5_cache: e["REF_PATH"] = os.path.join(d, "md5", "%2s", "%2s", "%s")
    return e


def view(exe, args, d, out=None, ok=True, env=None):
    """run test_view in the fixture directory; -> stdout bytes (or writes `out`)"""
    p = subprocess.run([exe] + list(args), cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env or _env(d), timeout=600)
    if ok:
        assert p.returncode == 0, (exe, args, p.returncode, p.stderr.decode("latin1")[-1500:])
    if out:
        open(os.path.join(d, out), "wb").write(p.stdout)
    return p.stdout


def canon(sam_text):
    """SAM text with the optional fields of every record sorted: what the reference's own harness compares (test/compare_sam.pl reads the tags into a hash).  A CRAM
    writer may store MD:Z / NM or leave them to the decoder, which appends the regenerated ones behind the stored tags (cram_decode.c:1111-1137): the order of a
    record's tags is the one thing two valid writers legitimately differ in."""
    out = []
    for ln in sam_text.split(b"\n"):
        f = ln.split(b"\t")
        out.append(ln if ln.startswith(b"@") or len(f) < 12 else b"\t".join(f[:11] + sorted(f[11:])))
    return b"\n".join(out)


def sams(d):
    return sorted(f for f in os.listdir(d) if "#" in f and f.endswith(".sam"))


def at(n):
    return [f"-@{n}"] if n else []


# ------------------------------------------------------------------------------------------------------------ test/test_bgzf.c inside libhts
def test_reference_test_bgzf_on_libhts_gpu(tmp_path):
    if not os.path.exists("/dev/kfd"): pytest.skip("no GPU")
    dropin_cases.reference_test_bgzf(BGZF_GPU, str(tmp_path))


# ------------------------------------------------------------------------------------------------------------ BAM: bam_write1 / bam_read1 on our bgzf
@pytest.mark.parametrize("threads", [0, 4])
def test_view_bam_both_directions_equal_stock_htslib(fx, threads):
    bam_both_directions(fx, threads, VIEW_GPU)


def bam_both_directions(fx, threads, VIEW_GPU):
    """test.pl:750-762 (SAM -> BAM -> SAM, compressed and -l0) for every *#*.sam: what ref_view_gpu writes, stock ref_view reads, and the other way round --
    SAM text identical to stock -> stock."""
    t = at(threads)
    files = sams(fx) if threads == 0 else sorted(CRAM31 | CRAM_MS)
    for sam in files:
        for lv in ([], ["-l0"]) if sam in ("ce#1000.sam", "xx#large_aux.sam", "ce#5b.sam") else (["-l0"],) if threads == 0 else ([],):
            view(VIEW_REF, ["-S", "-b", *lv, sam], fx, "stock.bam")
            want = view(VIEW_REF, ["stock.bam"], fx)
            view(VIEW_GPU, [*t, "-S", "-b", *lv, sam], fx, "gpu.bam")
            assert view(VIEW_REF, ["gpu.bam"], fx) == want, (sam, lv, "stock reads what we wrote")
            assert view(VIEW_GPU, [*t, "stock.bam"], fx) == want, (sam, lv, "we read what stock wrote")
            if threads: assert view(VIEW_GPU, [*t, "gpu.bam"], fx) == want, (sam, lv, "round trip")


# ------------------------------------------------------------------------------------------------------------ CRAM <= 3.0: both directions against stock
@pytest.mark.parametrize("threads", [0, 4])
def test_view_cram30_both_directions_equal_stock_htslib(fx, threads):
    """test.pl:764-790: SAM -> CRAM 2.1 / 3.0 (+ multi-slice containers) -> SAM.  ref_view_gpu -C = the reference's cram_encode_slice + OUR
    cram_compress_block2 / cram_write_block; reading = the reference's cram_decode_slice + OUR cram_read_block / cram_uncompress_block."""
    t = at(threads)
    files = sams(fx) if threads == 0 else sorted(CRAM31 | CRAM_MS)
    for sam in files:
        ref = sam.split("#")[0] + ".fa"
        combos = [["-o", "VERSION=3.0"]]
        if threads == 0 or sam in ("ce#1000.sam", "ce#5b.sam"): combos.append(["-o", "VERSION=2.1"])
        if sam in CRAM_MS: combos.append(["-o", "VERSION=3.0", "-o", "seqs_per_slice=7", "-o", "slices_per_container=5"])
        for o in combos:
            view(VIEW_REF, ["-t", ref, "-S", "-C", *o, sam], fx, "stock.cram")
            want = view(VIEW_REF, ["-D", "stock.cram"], fx)
            view(VIEW_GPU, [*t, "-t", ref, "-S", "-C", *o, sam], fx, "gpu.cram")
            # (the writer under cram_put_bam_seq is OURS for the default options -- whole slices on the device, MD:Z / NM kept as stored -- and the reference's
            # cram_encode_slice on our block layer for the others: tag order is the writer's choice, everything else must be stock's)
            assert canon(view(VIEW_REF, ["-D", "gpu.cram"], fx)) == canon(want), (sam, o, "stock reads what we wrote")
            assert view(VIEW_GPU, [*t, "-D", "stock.cram"], fx) == want, (sam, o, "we read what stock wrote")
            if threads: assert canon(view(VIEW_GPU, [*t, "-D", "gpu.cram"], fx)) == canon(want), (sam, o, "round trip")


def test_view_reads_the_htsjdk_crams(fx):
    """test.pl:818-824: the pre-made Java CRAMs (rANS 4x8 order 0/1, the pinned vectors) through the reference's decoder on our block layer"""
    for base, ref in (("auxf#values", "auxf.fa"), ("ce#5b", "ce.fa"), ("xx#large_aux", "xx.fa")):
        want = view(VIEW_REF, ["-i", "reference=" + ref, base + "_java.cram"], fx)
        assert len(want) > 100
        for t in (0, 4):
            assert view(VIEW_GPU, [*at(t), "-i", "reference=" + ref, base + "_java.cram"], fx) == want, (base, t)
    r = ["-i", "reference=ce.fa"]
    want = view(VIEW_REF, [*r, "range.cram"], fx)
    assert len(want) > 10000 and view(VIEW_GPU, ["-@4", *r, "range.cram"], fx) == want
    assert view(VIEW_GPU, [*r, "range.cram", "CHROMOSOME_II:2000-3000"], fx) == view(VIEW_REF, [*r, "range.cram", "CHROMOSOME_II:2000-3000"], fx)


# ------------------------------------------------------------------------------------------------------------ CRAM 3.1: every method family on the device
@pytest.mark.parametrize("threads", [0, 4])
def test_view_cram31_profiles_round_trip_and_cross_check(fx, threads):
    """test.pl:792-803: SAM -> CRAM 3.1 {fast, normal, small, archive} -> SAM (-l7).  The writer is the reference's cram_compress_slice with its real method
    sets (rANS Nx16, range coder, fqzcomp, tok3 ...) on OUR codecs; stock htslib has no 3.1 codec in this container (htscodecs is an absent submodule), so
    the bar is the reference's own: a self round trip, equal to the 3.0 text -- plus the independent restatements under oracle/ (ORC_STUB_CODECS31=1)
    reading the same files."""
    t = at(threads)
    e31 = dict(_env(fx), ORC_STUB_CODECS31="1")
    for sam in sorted(CRAM31):
        ref = sam.split("#")[0] + ".fa"
        profiles = ["fast", "normal", "small", "archive"] if sam == "ce#1000.sam" else ["archive"]
        view(VIEW_REF, ["-t", ref, "-S", "-l7", "-C", "-o", "VERSION=3.0", sam], fx, "stock30.cram")
        want = view(VIEW_REF, ["-D", "stock30.cram"], fx)
        for prof in profiles:
            view(VIEW_GPU, [*t, "-t", ref, "-S", "-l7", "-C", "-o", "VERSION=3.1", "-o", prof, sam], fx, "gpu31.cram")
            assert open(os.path.join(fx, "gpu31.cram"), "rb").read(6) == b"CRAM\x03\x01"
            assert canon(view(VIEW_GPU, [*t, "-D", "gpu31.cram"], fx)) == canon(want), (sam, prof, "round trip")
            assert canon(view(VIEW_REF, ["-D", "gpu31.cram"], fx, env=e31)) == canon(want), (sam, prof, "the CPU restatements read what the device wrote")
            view(VIEW_REF, ["-t", ref, "-S", "-l7", "-C", "-o", "VERSION=3.1", "-o", prof, sam], fx, "orc31.cram", env=e31)
            assert view(VIEW_GPU, [*t, "-D", "orc31.cram"], fx) == want, (sam, prof, "the device reads what the CPU restatements wrote")


def test_cram31_files_use_the_31_methods(fx):
    """the archive profile of ce#1000 must really contain method 5-8 blocks (not a gzip fall-back): count on-disk method ids through the reference's reader"""
    view(VIEW_GPU, ["-@4", "-t", "ce.fa", "-S", "-l7", "-C", "-o", "VERSION=3.1", "-o", "archive", "ce#1000.sam"], fx, "m31.cram")
    import sys
    sys.path.insert(0, os.path.join(refutil.ROOT, "tests", "golden"))
    import make_golden_rans as R                                 # (its walker of the CRAM container format; test infrastructure)
    meth = {blk[0] for _, blks in R.containers(open(os.path.join(fx, "m31.cram"), "rb").read()) for blk in blks}
    assert any(m in meth for m in (5, 6)) and 8 in meth, meth


# ------------------------------------------------------------------------------------------------------------ test.pl test_index
def _cmp(d, got, want, gz=False):
    a, b = open(os.path.join(d, got), "rb").read(), open(os.path.join(d, want), "rb").read()
    if gz: a, b = gzip.decompress(a), gzip.decompress(b)
    assert a == b, (got, want, len(a), len(b))


@pytest.mark.parametrize("threads", [0, 4])
def test_index_on_the_fly_and_test_index_equal_golden(fx, threads):
    index_scenarios(fx, threads, VIEW_GPU, INDEX_GPU)


def index_scenarios(fx, threads, VIEW_GPU, INDEX_GPU):
    """test.pl:1067-1160 with test_view / test_index on libhts_gpu.so; the golden indexes are the reference's own files."""
    t = at(threads)
    d = fx
    env5 = _env(d, md5_cache=True)

    def idx(args):
        p = subprocess.run([INDEX_GPU] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env5, timeout=600)
        assert p.returncode == 0, (args, p.stderr.decode("latin1")[-1500:])

    def view(exe, args, d, out=None):                      # (REF_PATH = the MD5 cache throughout this scenario)
        return globals()["view"](exe, args, d, out, env=env5)

    def rm(f):
        if os.path.exists(os.path.join(d, f)): os.unlink(os.path.join(d, f))

    os.makedirs(os.path.join(d, "t"), exist_ok=True)
    # BAM
    view(VIEW_GPU, [*t, "-l", "0", "-b", "-m", "14", "-x", "t/index.bam.csi", "index.sam"], d, "t/index.bam"); _cmp(d, "t/index.bam.csi", "index.bam.csi", gz=True)
    rm("t/index.bam.csi"); idx(["-c", "t/index.bam"]); _cmp(d, "t/index.bam.csi", "index.bam.csi", gz=True)
    view(VIEW_GPU, [*t, "-l", "0", "-b", "-m", "0", "-x", "t/index.bam.bai", "index.sam"], d, "t/index.bam"); _cmp(d, "t/index.bam.bai", "index.bam.bai")
    rm("t/index.bam.bai"); idx(["-b", "t/index.bam"]); _cmp(d, "t/index.bam.bai", "index.bam.bai")
    # SAM.gz (and DOS line endings)
    for src in ("index.sam", "index_dos.sam"):
        view(VIEW_GPU, [*t, "-l", "0", "-z", "-m", "14", "-x", "t/index.sam.gz.csi", src], d, "t/index.sam.gz"); _cmp(d, "t/index.sam.gz.csi", "index.sam.gz.csi", gz=True)
        rm("t/index.sam.gz.csi"); idx(["-c", "t/index.sam.gz"]); _cmp(d, "t/index.sam.gz.csi", "index.sam.gz.csi", gz=True)
        view(VIEW_GPU, [*t, "-l", "0", "-z", "-m", "0", "-x", "t/index.sam.gz.bai", src], d, "t/index.sam.gz"); _cmp(d, "t/index.sam.gz.bai", "index.sam.gz.bai")
        rm("t/index.sam.gz.bai"); idx(["-b", "t/index.sam.gz"]); _cmp(d, "t/index.sam.gz.bai", "index.sam.gz.bai")
    # CRAM
    view(VIEW_GPU, [*t, "-l", "0", "-C", "-x", "t/index.cram.crai", "index.sam"], d, "t/index.cram"); _cmp(d, "t/index.cram.crai", "index.cram.crai", gz=True)
    rm("t/index.cram.crai"); idx(["t/index.cram"]); _cmp(d, "t/index.cram.crai", "index.cram.crai", gz=True)
    # CRAM container skipping
    view(VIEW_GPU, [*t, "-C", "-p", "t/index3.cram", "-x", "t/index3.cram.crai", "-o", "seqs_per_slice=2", "index3.sam"], d)
    view(VIEW_GPU, [*t, "-p", "t/index3_rgn.sam", "t/index3.cram", "CHROMOSOME_I:5000-5100"], d); _cmp(d, "t/index3_rgn.sam", "index3_exp.sam")
    view(VIEW_GPU, [*t, "-M", "-p", "t/index3_rgnm.sam", "t/index3.cram", "CHROMOSOME_I:5000-5100"], d); _cmp(d, "t/index3_rgnm.sam", "index3_exp.sam")
    # BCF / VCF
    view(VIEW_GPU, [*t, "-l", "0", "-b", "-m", "14", "-x", "t/index.bcf.csi", "index.vcf"], d, "t/index.bcf"); _cmp(d, "t/index.bcf.csi", "index.bcf.csi", gz=True)
    rm("t/index.bcf.csi"); idx(["-c", "t/index.bcf"]); _cmp(d, "t/index.bcf.csi", "index.bcf.csi", gz=True)
    view(VIEW_GPU, [*t, "-l", "0", "-z", "-m", "14", "-x", "t/index.vcf.gz.csi", "index.vcf"], d, "t/index.vcf.gz"); _cmp(d, "t/index.vcf.gz.csi", "index.vcf.gz.csi", gz=True)
    rm("t/index.vcf.gz.csi"); idx(["-c", "t/index.vcf.gz"]); _cmp(d, "t/index.vcf.gz.csi", "index.vcf.gz.csi", gz=True)
    view(VIEW_GPU, [*t, "-l", "0", "-z", "-m", "0", "-x", "t/index.vcf.gz.tbi", "index.vcf"], d, "t/index.vcf.gz"); _cmp(d, "t/index.vcf.gz.tbi", "index.vcf.gz.tbi", gz=True)
    rm("t/index.vcf.gz.tbi"); idx(["-t", "t/index.vcf.gz"]); _cmp(d, "t/index.vcf.gz.tbi", "index.vcf.gz.tbi", gz=True)
    # index2: mapped/unmapped pairs, every query returns exactly two records
    view(VIEW_GPU, ["-b", "-p", "t/index2.bam", "-x", "t/index2.bam.bai", "index2.sam"], d)
    for tid in (1, 2):
        for pos in (1, 2):
            out = view(VIEW_GPU, ["t/index2.bam", f"{tid}:{pos}000000-{pos}000000"], d)
            assert sum(1 for ln in out.splitlines() if ln and not ln.startswith(b"@")) == 2


def test_compressed_bam_with_index_on_the_fly_equals_stock(fx):
    """a COMPRESSED, threaded writer (deferred block addresses: bgzf_idx_push resolves the virtual offsets when the device batch returns, bgzf.c:189-290) --
    the index must describe OUR file: region queries through it return what stock htslib returns from its own file + index."""
    view(VIEW_REF, ["-b", "-p", "t/stock.bam", "-x", "t/stock.bam.bai", "-m", "0", "ce#1000.sam"], fx)
    view(VIEW_GPU, ["-@4", "-b", "-p", "t/gpu.bam", "-x", "t/gpu.bam.bai", "-m", "0", "ce#1000.sam"], fx)
    view(VIEW_GPU, ["-@4", "-b", "-p", "t/gpu2.bam", "-x", "t/gpu2.bam.csi", "-m", "14", "xx#large_aux.sam"], fx)
    for rgn in ("CHROMOSOME_I:1-1000", "CHROMOSOME_I:900-1100", "CHROMOSOME_I:1", "CHROMOSOME_I:100000"):
        want = view(VIEW_REF, ["t/stock.bam", rgn], fx)
        assert view(VIEW_GPU, ["t/gpu.bam", rgn], fx) == want, rgn
        assert view(VIEW_REF, ["t/gpu.bam", rgn], fx) == want, rgn          # stock htslib + our file + our index
    want = view(VIEW_REF, ["-S", "-b", "xx#large_aux.sam"], fx, "t/stock2.bam")
    assert view(VIEW_REF, ["t/gpu2.bam", "xx"], fx) == view(VIEW_GPU, ["t/gpu2.bam", "xx"], fx)
    assert view(VIEW_REF, ["t/gpu2.bam"], fx) == view(VIEW_REF, ["t/stock2.bam"], fx)


# ------------------------------------------------------------------------------------------------------------ the whole-slice reader under cram_get_bam_seq
def _reader_stats(stderr):
    return [ln for ln in stderr.decode("latin1").splitlines() if "cram reader:" in ln or "cram run" in ln]


def test_whole_slice_reader_is_the_path_that_runs_and_its_switch_is_honoured(fx):
    """cram_get_bam_seq inside libhts_gpu.so = htslib_amd/csrc/cram_record_front.c (reference cram/cram_decode.c:3615): a plain sequential read goes through runs of
    containers on the device (fused: blocks decoded in HBM, the record decoder beside them); HTS_GPU_CRAM_SLICE=0 leaves the reference's cram_decode_slice on our
    per-block entry points; a region query (fd->range set) is the reference's reader by design.  All three print the SAM text stock htslib prints."""
    view(VIEW_REF, ["-t", "ce.fa", "-S", "-C", "-o", "VERSION=3.0", "-o", "seqs_per_slice=100", "ce#1000.sam"], fx, "rd.cram")
    want = view(VIEW_REF, ["-D", "rd.cram"], fx)
    e = dict(_env(fx), HTS_GPU_STATS="1")
    p = subprocess.run([VIEW_GPU, "-D", "rd.cram"], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
    assert p.returncode == 0 and p.stdout == want
    st = _reader_stats(p.stderr)
    assert any("cram run (fused): 10 containers, 10 slices" in ln and "(rc 0)" in ln for ln in st), st
    assert any("cram reader:" in ln and "1000 records" in ln for ln in st), st
    p = subprocess.run([VIEW_GPU, "-D", "rd.cram"], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(e, HTS_GPU_CRAM_SLICE="0"), timeout=600)
    assert p.returncode == 0 and p.stdout == want and not _reader_stats(p.stderr)
    p = subprocess.run([VIEW_GPU, "-D", "rd.cram"], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(e, HTS_GPU_CRAM_FUSED="0"), timeout=600)
    assert p.returncode == 0 and p.stdout == want and any("cram run:" in ln for ln in _reader_stats(p.stderr))        # the host-buffer composition of the same run


def test_whole_slice_reader_hands_the_hard_cases_to_the_right_path(fx):
    """embedded references and reference-less files (the run decoder digests an embedded block on the host: not fused), multi-reference slices (references asked for
    by the ids in the RI series), multi-slice containers, an unsorted file: each equal to stock, each through the reader (no silent fall-back: the stats line
    must show every record coming out of it)."""
    cases = [("ce#1000.sam", "ce.fa", ["-o", "embed_ref=1"]), ("ce#1000.sam", "ce.fa", ["-o", "no_ref=1"]), ("ce#5b.sam", "ce.fa", ["-o", "multi_seq_per_slice=1"]),
             ("ce#unmap2.sam", "ce.fa", ["-o", "seqs_per_slice=7", "-o", "slices_per_container=5"]), ("xx#unsorted.sam", "xx.fa", [])]
    e = dict(_env(fx), HTS_GPU_STATS="1")
    for sam, ref, o in cases:
        view(VIEW_REF, ["-t", ref, "-S", "-C", "-o", "VERSION=3.0", *o, sam], fx, "hard.cram")
        want = view(VIEW_REF, ["-D", "hard.cram"], fx)
        nrec = sum(1 for ln in want.splitlines() if ln and not ln.startswith(b"@"))
        p = subprocess.run([VIEW_GPU, "-D", "hard.cram"], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
        assert p.returncode == 0 and p.stdout == want, (sam, o, p.stderr.decode("latin1")[-800:])
        assert any("cram reader:" in ln and f" {nrec} records" in ln for ln in _reader_stats(p.stderr)), (sam, o, _reader_stats(p.stderr))


def test_whole_slice_reader_leaves_damaged_files_to_the_reference(fx):
    """a truncated file, a flipped payload byte (block CRC), a wrong reference (slice MD5): the run that meets the damage is handed back WHOLE to the reference's
    reader (seek to its first container), so the records before it, the error text's last line and the exit status are stock htslib's."""
    view(VIEW_REF, ["-t", "ce.fa", "-S", "-C", "-o", "VERSION=3.0", "-o", "seqs_per_slice=100", "ce#1000.sam"], fx, "dmg.cram")
    raw = open(os.path.join(fx, "dmg.cram"), "rb").read()
    flip = bytearray(raw); flip[len(raw) * 2 // 3] ^= 0x40
    open(os.path.join(fx, "dmg_trunc.cram"), "wb").write(raw[:len(raw) * 3 // 5])
    open(os.path.join(fx, "dmg_flip.cram"), "wb").write(bytes(flip))
    fa = open(os.path.join(fx, "ce.fa")).read().replace("GCCTAAGCC", "GCCTTAGCC")
    open(os.path.join(fx, "ce_wrong.fa"), "w").write(fa)
    subprocess.run([VIEW_REF, "-t", "ce_wrong.fa", "-S", "-b", "ce#5.sam"], cwd=fx, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=_env(fx))      # (writes ce_wrong.fa.fai)
    for f, args in (("dmg_trunc.cram", []), ("dmg_flip.cram", []), ("dmg.cram", ["-i", "reference=ce_wrong.fa"])):
        r = subprocess.run([VIEW_REF, *args, f], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(fx), timeout=600)
        g = subprocess.run([VIEW_GPU, *args, f], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(fx), timeout=600)
        assert r.returncode != 0, (f, "the damage must be one stock htslib notices")
        assert g.returncode == r.returncode and g.stdout == r.stdout, (f, r.returncode, g.returncode, len(r.stdout), len(g.stdout), g.stderr.decode("latin1")[-600:])
        assert g.stderr.decode("latin1").strip().splitlines()[-1:] == r.stderr.decode("latin1").strip().splitlines()[-1:], (f, r.stderr[-300:], g.stderr[-300:])


# ------------------------------------------------------------------------------------------------------------ the whole-slice writer under cram_put_bam_seq
def test_whole_slice_writer_is_the_path_that_runs_and_hands_over_what_it_cannot_write(fx):
    """cram_put_bam_seq inside libhts_gpu.so (cram_record_front.c; reference cram/cram_encode.c:4042): default options -> runs of records through the device writer
    (hg_cram_writer_containers_host), stock htslib reads the file back to the input; options this writer does not implement (here: several slices per container,
    an embedded reference) -> the reference's writer from the first record; a run it cannot encode (no reference to be had: the reference's writer then embeds one
    by itself) -> that run is replayed through the reference's writer.  In every case the SAM text is what stock htslib's own file gives (tags in any order)."""
    e = dict(_env(fx), HTS_GPU_STATS="1")

    def write(args, env=e):
        p = subprocess.run([VIEW_GPU, *args], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        assert p.returncode == 0, (args, p.stderr.decode("latin1")[-800:])
        open(os.path.join(fx, "w.cram"), "wb").write(p.stdout)
        return [ln for ln in p.stderr.decode("latin1").splitlines() if "cram writer:" in ln or "bam_to_cram:" in ln]

    view(VIEW_REF, ["-t", "ce.fa", "-S", "-C", "-o", "VERSION=3.0", "ce#1000.sam"], fx, "s.cram")
    want = canon(view(VIEW_REF, ["-D", "s.cram"], fx))
    st = write(["-t", "ce.fa", "-S", "-C", "-o", "VERSION=3.0", "-o", "seqs_per_slice=100", "ce#1000.sam"])
    assert any("bam_to_cram: 1000 records, 10 slices" in ln for ln in st) and any("cram writer: 1 runs, 1000 records through the device;" in ln for ln in st), st
    assert canon(view(VIEW_REF, ["-D", "w.cram"], fx)) == want
    for o in (["-o", "slices_per_container=3"], ["-o", "embed_ref=1"]):
        st = write(["-t", "ce.fa", "-S", "-C", "-o", "VERSION=3.0", *o, "ce#1000.sam"])
        assert not st, (o, st)                                               # never started: the reference's writer on our block layer
        assert canon(view(VIEW_REF, ["-D", "w.cram"], fx)) == want, o
    assert not write(["-t", "ce.fa", "-S", "-C", "-o", "VERSION=3.0", "ce#1000.sam"], env=dict(e, HTS_GPU_CRAM_SLICE="0"))
    # no reference file and none to be found: whichever way it goes (never started, or the first run handed back and replayed), the reference's writer wrote the file
    st = write(["-S", "-C", "-o", "VERSION=3.0", "ce#5.sam"])
    assert not any("cram writer:" in ln and " 0 records through the device" not in ln for ln in st), st
    view(VIEW_REF, ["-S", "-C", "-o", "VERSION=3.0", "ce#5.sam"], fx, "s5.cram")
    assert canon(view(VIEW_REF, ["-D", "w.cram"], fx)) == canon(view(VIEW_REF, ["-D", "s5.cram"], fx))


def test_reader_and_writer_in_one_process_transcode_cram(fx):
    """`test_view -C in.cram` / `-b in.cram`: the whole-slice reader feeds the whole-slice writer (or bam_write1 on our bgzf_write) in the same process, both on the block layer's
    device context; stock htslib reads the result back to what it reads from the input."""
    view(VIEW_REF, ["-t", "ce.fa", "-S", "-C", "-o", "VERSION=3.0", "-o", "seqs_per_slice=100", "ce#1000.sam"], fx, "in.cram")
    want = canon(view(VIEW_REF, ["-D", "in.cram"], fx))
    e = dict(_env(fx), HTS_GPU_STATS="1")
    p = subprocess.run([VIEW_GPU, "-@4", "-D", "-t", "ce.fa", "-C", "-o", "VERSION=3.0", "in.cram"], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
    assert p.returncode == 0, p.stderr.decode("latin1")[-800:]
    st = p.stderr.decode("latin1")
    assert "cram reader: " in st and "1000 records" in st and "cram writer: 1 runs, 1000 records through the device;" in st, st[-1500:]
    open(os.path.join(fx, "tc.cram"), "wb").write(p.stdout)
    assert canon(view(VIEW_REF, ["-D", "tc.cram"], fx)) == want
    view(VIEW_GPU, ["-@4", "-D", "-b", "in.cram"], fx, "tc.bam")
    assert canon(view(VIEW_REF, ["tc.bam"], fx)) == want


def test_whole_slice_reader_takes_cram31_and_multi_reference_slices_through_the_fused_run_decoder(fx):
    """CRAM 3.1 (the reference's default output version): rANS Nx16 / tok3 blocks are decoded inside the fused run decoder (hg_entropy_decode_inplace_*, tok3 beside it), and a
    multi-reference slice (its RI block fetched from the device) no longer sends the run through the host-buffer composition.  Checker: the file read back through the
    CPU restatements of the 3.1 codecs (ORC_STUB_CODECS31=1) and through the reference's decoder on our per-block layer (HTS_GPU_CRAM_SLICE=0)."""
    e = dict(_env(fx), HTS_GPU_STATS="1")
    e31 = dict(_env(fx), ORC_STUB_CODECS31="1")
    for sam, ref, o in (("ce#1000.sam", "ce.fa", ["-o", "seqs_per_slice=100"]), ("ce#5b.sam", "ce.fa", ["-o", "multi_seq_per_slice=1"])):
        view(VIEW_REF, ["-t", ref, "-S", "-C", "-o", "VERSION=3.1", *o, sam], fx, "v31.cram", env=e31)
        want = view(VIEW_REF, ["-D", "v31.cram"], fx, env=e31)
        assert open(os.path.join(fx, "v31.cram"), "rb").read(6) == b"CRAM\x03\x01"
        p = subprocess.run([VIEW_GPU, "-D", "v31.cram"], cwd=fx, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
        assert p.returncode == 0 and p.stdout == want, (sam, p.stderr.decode("latin1")[-800:])
        st = _reader_stats(p.stderr)
        assert any("cram run (fused):" in ln and "(rc 0)" in ln for ln in st) and not any("cram run:" in ln for ln in st), (sam, st)
        assert view(VIEW_GPU, ["-D", "v31.cram"], fx, env=dict(_env(fx), HTS_GPU_CRAM_SLICE="0")) == want
