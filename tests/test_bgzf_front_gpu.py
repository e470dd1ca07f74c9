"""GPU tests of the BGZF front-end (include/hts_bgzf_gpu.h), modelled on the reference's
test/test_bgzf.c: read the shipped fixture, write/read round trips at several modes, tell/seek,
getc/peek/getline, EOF-marker handling, .gzi index, interoperability with the real reference."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from tests import refutil, bgzf_capi
from htslib_amd import synth

pytestmark = pytest.mark.gpu
GOLD = refutil.GOLDEN


@pytest.fixture(scope="module")
def L(built, engine):
    return bgzf_capi.load()


def write_file(L, path, data, mode=b"w", pieces=None):
    fp = L.bgzf_open(path.encode(), mode)
    assert fp
    if pieces is None:
        assert L.bgzf_write(fp, data, len(data)) == len(data)
    else:
        pos = 0
        for n in pieces:
            assert L.bgzf_flush_try(fp, n) == 0
            assert L.bgzf_write(fp, data[pos:pos + n], n) == n
            pos += n
    assert L.bgzf_close(fp) == 0


def test_read_reference_fixture(L):                               # test_bgzf.c test_read (:403)
    fp = L.bgzf_open(os.path.join(GOLD, "bgziptest.txt.gz").encode(), b"r")
    assert fp and fp.contents.is_compressed == 1 and fp.contents.is_write == 0
    want = open(os.path.join(GOLD, "bgziptest.txt.gz.plain"), "rb").read()
    assert bgzf_capi.read_all(L, fp, 7) == want
    assert fp.contents.errcode == 0 and fp.contents.no_eof_block == 0
    assert L.bgzf_close(fp) == 0


@pytest.mark.parametrize("mode", [b"w", b"w0", b"w1", b"w9", b"wu"])
def test_write_read_modes_and_interop(L, tmp_path, mode):         # test_bgzf.c :1089-1100
    data = synth.fastq(700_000)
    p = str(tmp_path / "t.gz")
    write_file(L, p, data, mode)
    raw = open(p, "rb").read()
    if mode == b"wu":
        assert raw == data
    else:
        assert raw[-28:] == synth.BGZF_EOF
        assert all(b[2] <= 0xFF00 for b in refutil.split_blocks(raw))
        if refutil.have_ref():
            assert refutil.ref_bgzip(["-d"], raw, "libdeflate") == data      # stock htslib reads our file
    fp = L.bgzf_open(p.encode(), b"r")
    assert bgzf_capi.read_all(L, fp) == data
    assert L.bgzf_close(fp) == 0
    if refutil.have_ref() and mode != b"wu":                                  # we read stock htslib's file
        q = str(tmp_path / "ref.gz")
        open(q, "wb").write(refutil.ref_bgzip(["-l", "6"], data, "zlib"))
        fp = L.bgzf_open(q.encode(), b"r")
        assert bgzf_capi.read_all(L, fp, 33333) == data
        L.bgzf_close(fp)


def test_tell_seek_getline_getc_peek(L, tmp_path):                 # test_bgzf.c :1117-1155
    lines = [b"line %d " % i + b"x" * (i % 97) for i in range(20000)]
    data = b"\r\n".join(lines) + b"\r\n"
    p = str(tmp_path / "lines.gz")
    write_file(L, p, data)
    fp = L.bgzf_open(p.encode(), b"r")
    ks = bgzf_capi.KString(0, 0, None)
    marks = []
    for i in range(len(lines)):
        if i % 997 == 0:
            marks.append((i, bgzf_capi.tell(fp)))
        n = L.bgzf_getline(fp, ord("\n"), C.byref(ks))
        assert n == len(lines[i]) and C.string_at(ks.s, ks.l) == lines[i]      # trailing \r stripped
    assert L.bgzf_getline(fp, ord("\n"), C.byref(ks)) == -1                     # EOF
    for i, v in reversed(marks):
        assert L.bgzf_seek(fp, v, 0) == 0
        assert bgzf_capi.tell(fp) == v
        assert L.bgzf_peek(fp) == lines[i][0]
        assert L.bgzf_getc(fp) == lines[i][0]
        n = L.bgzf_getline(fp, ord("\n"), C.byref(ks))
        assert C.string_at(ks.s, ks.l) == lines[i][1:]
    assert L.bgzf_seek(fp, 0, 1) == -1 and fp.contents.errcode & 8            # SEEK_CUR is misuse
    L.bgzf_close(fp)
    C.CDLL(None).free(C.c_void_p(ks.s))


def test_eof_marker_handling(L, tmp_path, capfd):                   # test_bgzf.c :1109-1111, bgzf.c:1047-1050
    data = synth.fastq(200_000)
    p = str(tmp_path / "a.gz")
    write_file(L, p, data[:100_000])
    write_file(L, p, data[100_000:], b"a")                          # append: embedded EOF block mid-file
    raw = open(p, "rb").read()
    assert raw.count(synth.BGZF_EOF) >= 2
    fp = L.bgzf_open(p.encode(), b"r")
    assert L.bgzf_check_EOF(fp) == 1
    assert bgzf_capi.read_all(L, fp) == data and fp.contents.no_eof_block == 0
    L.bgzf_close(fp)
    q = str(tmp_path / "noeof.gz")
    open(q, "wb").write(raw[:-28])
    fp = L.bgzf_open(q.encode(), b"r")
    assert L.bgzf_check_EOF(fp) == 0
    assert bgzf_capi.read_all(L, fp) == data                        # a missing marker is a warning, not an error
    assert fp.contents.no_eof_block == 1 and fp.contents.errcode == 0
    L.bgzf_close(fp)


def test_flush_try_keeps_records_whole(L, tmp_path):                # sam.c:888, bgzf.c:1996-2000
    data, starts, hdr_len = synth.bam_stream(1 << 20)
    sizes = [hdr_len] + np.diff(np.append(starts, len(data))).tolist()
    p = str(tmp_path / "rec.bam")
    fp = L.bgzf_open(p.encode(), b"w")
    pos = 0
    for k, n in enumerate(sizes):
        assert L.bgzf_flush_try(fp, n) == 0
        assert L.bgzf_write(fp, data[pos:pos + n], n) == n
        if k == 0:
            assert L.bgzf_flush(fp) == 0                             # bam_hdr_write flushes the header
        pos += n
    assert L.bgzf_close(fp) == 0
    raw = open(p, "rb").read()
    cuts = np.concatenate([[0], np.cumsum([b[2] for b in refutil.split_blocks(raw)])])
    assert cuts.tolist() == synth.cut_blocks(len(data), starts, hdr_len).tolist() + [len(data)]   # + EOF block
    fp = L.bgzf_open(p.encode(), b"r")
    assert bgzf_capi.read_all(L, fp) == data
    L.bgzf_close(fp)


def test_gzi_index_matches_reference_and_useek(L, tmp_path):        # test_bgzf.c index tests, bgzf.c:2385-2542
    src = os.path.join(GOLD, "bgziptest.txt.gz")
    fp = L.bgzf_open(src.encode(), b"r")
    assert L.bgzf_index_build_init(fp) == 0
    bgzf_capi.read_all(L, fp)
    out = str(tmp_path / "x")
    assert L.bgzf_index_dump(fp, out.encode(), b".gzi") == 0
    L.bgzf_close(fp)
    # An index made while READING lists the start of every data block but the first (bgzf.c:1066-1076, 2385-2411).  The
    # shipped bgziptest.txt.gz.gzi additionally lists the EOF block; htslib 1.23's own reader (`bgzip -r`) does not:
    got = open(out + ".gzi", "rb").read()
    fixture = open(src + ".gzi", "rb").read()
    assert got[8:] == fixture[8:-16] and struct.unpack_from("<Q", got)[0] == struct.unpack_from("<Q", fixture)[0] - 1
    if refutil.have_ref():
        import shutil, subprocess
        cp = str(tmp_path / "r.gz")
        shutil.copy(src, cp)
        subprocess.run([os.path.join(refutil.REF_DIR, "ref_bgzip"), "-r", cp], check=True)
        assert open(cp + ".gzi", "rb").read() == got                  # byte-identical to the real reference's reindex
    # index built while WRITING, then used for uncompressed-offset seeks
    data = synth.fastq(500_000)
    p = str(tmp_path / "w.gz")
    fp = L.bgzf_open(p.encode(), b"w")
    assert L.bgzf_index_build_init(fp) == 0
    assert L.bgzf_write(fp, data, len(data)) == len(data)
    assert L.bgzf_index_dump(fp, p.encode(), b".gzi") == 0
    assert L.bgzf_close(fp) == 0
    raw = open(p, "rb").read()
    blocks = refutil.split_blocks(raw)
    gzi = open(p + ".gzi", "rb").read()
    n = struct.unpack_from("<Q", gzi)[0]
    ent = [struct.unpack_from("<QQ", gzi, 8 + 16 * i) for i in range(n)]
    # one entry per data block after the first; the EOF block is not listed (bgzf.c:2354-2366, 2393-2395)
    assert ent == [(b[0], sum(x[2] for x in blocks[:i + 1])) for i, b in enumerate(blocks[1:-1])]
    fp = L.bgzf_open(p.encode(), b"r")
    assert L.bgzf_index_load(fp, p.encode(), b".gzi") == 0
    buf = C.create_string_buffer(100)
    for off in (0, 1, 65279, 65280, 65281, 300_000, len(data) - 50):
        assert L.bgzf_useek(fp, off, 0) == 0 and L.bgzf_utell(fp) == off
        assert L.bgzf_read(fp, buf, 50) == 50 and buf.raw[:50] == data[off:off + 50]
    L.bgzf_close(fp)


def test_single_block_bgzf_compress_and_errors(L, tmp_path, oracle):  # bgzf.c:561-683, 730-804
    data = synth.fastq(60_000)[:0xFF00]
    dst = C.create_string_buffer(65536)
    dlen = C.c_size_t(65536)
    assert L.bgzf_compress(dst, C.byref(dlen), data, len(data), 6) == 0
    rc, got = oracle.uncompress_block(dst.raw[:dlen.value])
    assert rc == 0 and got == data
    dlen = C.c_size_t(65536)
    assert L.bgzf_compress(dst, C.byref(dlen), b"", 0, 6) == 0 and dst.raw[:dlen.value] == synth.BGZF_EOF
    dlen = C.c_size_t(10)
    assert L.bgzf_compress(dst, C.byref(dlen), data, len(data), 6) == -1
    # CRC damage surfaces as BGZF_ERR_CRC from bgzf_read
    p = str(tmp_path / "bad.gz")
    good = synth.bgzf_compress(synth.fastq(300_000))
    blocks = refutil.split_blocks(good)
    off, clen, _ = blocks[2]
    bad = bytearray(good); bad[off + clen - 8] ^= 0xFF
    open(p, "wb").write(bytes(bad))
    fp = L.bgzf_open(p.encode(), b"r")
    buf = C.create_string_buffer(1 << 20)
    total = 0
    while True:
        n = L.bgzf_read(fp, buf, 4080)                              # 4080 divides 0xff00: reads end on block edges
        if n <= 0:
            break
        total += n
    assert n == -1 and fp.contents.errcode & 32                     # BGZF_ERR_CRC
    assert total == blocks[0][2] + blocks[1][2]                     # the good blocks before it were delivered
    # (like the reference, a read that runs into the bad block returns -1 even if it had copied bytes,
    #  bgzf.c:1262-1266)
    L.bgzf_close(fp)
    assert L.bgzf_is_bgzf(p.encode()) == 1
    assert not L.bgzf_open(str(tmp_path / "missing").encode(), b"r")


@pytest.mark.gpu
def test_dopen_reads_from_where_the_descriptor_stands(L, tmp_path):     # hfile.c hdopen: logical offset 0 = the descriptor's position
    """a descriptor that was positioned behind a foreign prefix (or behind the first of two concatenated BGZF streams) is read from THERE, as the
    reference's hread path does -- not from file offset 0 by the positional window reads"""
    rnd = np.random.default_rng(3)
    one = bytes(rnd.integers(65, 70, 300_000, dtype=np.uint8)); two = bytes(rnd.integers(70, 75, 500_000, dtype=np.uint8))
    p1, p2 = str(tmp_path / "one.gz"), str(tmp_path / "two.gz")
    write_file(L, p1, one); write_file(L, p2, two)
    prefix = b"#!not bgzf at all\n" * 1000
    cat = str(tmp_path / "cat.bin")
    open(cat, "wb").write(prefix + open(p1, "rb").read() + open(p2, "rb").read())
    for start, want in ((len(prefix), one + two), (len(prefix) + os.path.getsize(p1), two), (0, None)):
        fd = os.open(cat, os.O_RDONLY)
        os.lseek(fd, start, os.SEEK_SET)
        fp = L.bgzf_dopen(fd, b"r")
        assert fp
        if want is None:
            assert fp.contents.is_compressed == 0                        # offset 0 holds the text prefix: read through as plain bytes
            assert bgzf_capi.read_all(L, fp, 1 << 20)[:len(prefix)] == prefix
        else:
            assert bgzf_capi.read_all(L, fp) == want, start
        assert L.bgzf_close(fp) == 0
    fd = os.open(p2, os.O_RDONLY)                                        # the common case keeps the fast path: descriptor at the start of a regular file
    fp = L.bgzf_dopen(fd, b"r")
    assert bgzf_capi.read_all(L, fp) == two and L.bgzf_close(fp) == 0


@pytest.mark.gpu
def test_latency_path_runs_on_the_host_codec(L, tmp_path):            # bgzf.c:1004-1239 (ST reader), :2029-2060 (ST writer); SURVEY 8b
    """random access and unthreaded writing do not pay a device round trip per block: bgzf_seek + a 100-byte read decodes ONE block on the calling thread
    (the reference: ~0.1 ms; a device job: ~1.5 ms), a writer that never calls bgzf_mt() compresses each block as it is cut (a device job per block: 16 MB/s)
    and keeps bgzf_tell() exact; the bytes are the same as on the streaming path"""
    import time
    plain, _, _ = synth.bam_stream(48 << 20, 0x5EED0001, 0, True)
    p = str(tmp_path / "w.bam")
    fp = L.bgzf_open(p.encode(), b"w")                                  # no bgzf_mt(): the synchronous writer
    t0 = time.perf_counter()
    tells = []
    for at in range(0, len(plain), 50_000):
        piece = plain[at:at + 50_000]
        assert L.bgzf_write(fp, piece, len(piece)) == len(piece)
        tells.append(bgzf_capi.tell(fp))
    assert L.bgzf_close(fp) == 0
    dt = time.perf_counter() - t0
    rate = len(plain) / dt / 1e6
    assert tells == sorted(tells) and tells[-1] >> 16 > 0               # the block address moves while writing (exact bgzf_tell)
    comp = open(p, "rb").read()
    assert refutil.Oracle().decompress(comp)[1] == plain
    assert rate >= 60, rate                                              # 16 MB/s when every block was a device job of its own; ~200 MB/s here
    # random access: seek to block starts all over the file, read 100 bytes
    blocks = refutil.split_blocks(comp)
    uoffs = np.concatenate([[0], np.cumsum([b[2] for b in blocks])])
    fp = L.bgzf_open(p.encode(), b"r")
    buf = C.create_string_buffer(100)
    rng = np.random.default_rng(1)
    lat = []
    for k in rng.integers(0, len(blocks) - 1, 300):
        off = blocks[int(k)][0]
        t = time.perf_counter()
        assert L.bgzf_seek(fp, off << 16 | 17, 0) == 0 and L.bgzf_read(fp, buf, 100) == 100
        lat.append(time.perf_counter() - t)
        u = int(uoffs[int(k)]) + 17
        assert buf.raw == plain[u:u + 100]
    assert L.bgzf_close(fp) == 0
    med = sorted(lat)[len(lat) // 2] * 1e3
    print("non-mt writer %.0f MB/s, seek + 100 B read median %.3f ms" % (rate, med))
    assert med <= 0.5, med                                               # 1.6 ms through a device job; the reference ~0.11 ms
    # sequential reading across the host-decoded first blocks into device batches gives the same bytes
    fp = L.bgzf_open(p.encode(), b"r")
    assert bgzf_capi.read_all(L, fp, 1 << 20) == plain and L.bgzf_close(fp) == 0
