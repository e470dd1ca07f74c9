"""Scenarios shared by tests/test_front_host_logic.py (zlib test double, CPU) and tests/test_reference_programs.py
(the real gfx950 library): the reference's OWN programs -- test/test_bgzf.c and bgzip.c compiled unmodified -- driven
the way the reference's test harness drives them (Makefile:700 `test/test_bgzf test/bgziptest.txt`,
test/test.pl:449-640 test_bgzip, :1238-1260 test_rebgzip)."""
from __future__ import annotations

import os
import shutil
import subprocess

from tests import refutil

GOLD = refutil.GOLDEN


def text_corpus(n: int = 1_500_000) -> bytes:
    """FASTA-like text (the role of test/ce.fa in test.pl): lines, several BGZF blocks."""
    import numpy as np
    rng = np.random.default_rng(0x5EED0001)
    out = []
    size = 0
    i = 0
    while size < n:
        hdr = f">chr{i} synthetic sequence {i}\n".encode()
        body = rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=60 * 200, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04])
        lines = b"\n".join(bytes(body[k:k + 60]) for k in range(0, len(body), 60)) + b"\n"
        out += [hdr, lines]
        size += len(hdr) + len(lines)
        i += 1
    return b"".join(out)[:n]


def run(cmd, stdin=None, cwd=None, ok=(0,)):
    p = subprocess.run(cmd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=cwd, timeout=900)
    assert p.returncode in ok, f"{cmd} -> rc {p.returncode}\n{p.stderr.decode(errors='replace')[-2000:]}"
    return p.stdout


def reference_test_bgzf(exe: str, tmp: str):
    """`test_bgzf <source file>`: the program writes its temporary files next to the source, so it runs on a copy."""
    src = os.path.join(tmp, "bgziptest.txt")
    open(src, "wb").write(open(os.path.join(GOLD, "bgziptest.txt.gz.plain"), "rb").read())
    shutil.copy(os.path.join(GOLD, "bgziptest.txt.gz"), src + ".gz")
    shutil.copy(os.path.join(GOLD, "bgziptest.txt.gz.gzi"), src + ".gz.gzi")
    p = subprocess.run([exe, src], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=tmp, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]


def reference_bgzip(exe: str, tmp: str, threads: int, checker: str | None):
    """test.pl test_bgzip ($at = '' or '-@4') + test_rebgzip; `checker` = a stock-htslib bgzip (oracle/_ref) or None."""
    at = [f"-@{threads}"] if threads else []
    data = text_corpus()
    src = os.path.join(tmp, "ce.fa")
    open(src, "wb").write(data)
    comp, idx = os.path.join(tmp, "c.gz"), os.path.join(tmp, "c.gzi")
    # round trip with an index written on the fly
    open(comp, "wb").write(run([exe, *at, "-i", "-I", idx], stdin=data))
    assert run([exe, *at, "-d"], stdin=open(comp, "rb").read()) == data
    if checker:
        assert run([checker, "-d"], stdin=open(comp, "rb").read()) == data          # stock htslib reads our file
        ref_c, ref_i = os.path.join(tmp, "r.gz"), os.path.join(tmp, "r.gzi")
        open(ref_c, "wb").write(run([checker, "-i", "-I", ref_i], stdin=data))
        assert run([exe, *at, "-d"], stdin=open(ref_c, "rb").read()) == data        # and we read stock htslib's
        # same block cuts => the index of OUR file must locate the same uncompressed offsets
        def uaddrs(path):
            import struct
            raw = open(path, "rb").read()
            n = struct.unpack_from("<Q", raw)[0]
            assert len(raw) == 8 + 16 * n
            return [struct.unpack_from("<Q", raw, 16 + 16 * k)[0] for k in range(n)]
        assert uaddrs(idx) == uaddrs(ref_i), "uncompressed offsets of the .gzi differ"
    # --binary: blocks are not cut at line ends
    open(comp + ".b", "wb").write(run([exe, *at, "--binary", "-i", "-I", idx + ".b"], stdin=data))
    assert run([exe, *at, "-d"], stdin=open(comp + ".b", "rb").read()) == data
    # -b OFFSET (uses <file>.gzi) and -b OFFSET -I index on a copy
    shutil.copy(idx, comp + ".gzi")
    off = len(data) // 3
    assert run([exe, *at, "-b", str(off), "-d", comp]) == data[off:]
    assert run([exe, *at, "-b", str(off), "-s", "1000", "-d", comp]) == data[off:off + 1000]
    copy = os.path.join(tmp, "copy.gz")
    shutil.copy(comp, copy)
    assert run([exe, *at, "-b", str(off), "-d", "-I", idx, copy]) == data[off:]
    # multiple files, in place, then back
    a, b = os.path.join(tmp, "m1.txt"), os.path.join(tmp, "m2.txt")
    open(a, "wb").write(data[:400_000]); open(b, "wb").write(data[400_000:900_000])
    run([exe, *at, a, b])
    assert not os.path.exists(a) and os.path.exists(a + ".gz") and os.path.exists(b + ".gz")
    run([exe, *at, "-d", a + ".gz", b + ".gz"])
    assert open(a, "rb").read() == data[:400_000] and open(b, "rb").read() == data[400_000:900_000]
    # --output
    run([exe, *at, src, "-o", os.path.join(tmp, "o.gz")])
    run([exe, *at, "-d", os.path.join(tmp, "o.gz"), "--output", os.path.join(tmp, "o.txt")])
    assert open(os.path.join(tmp, "o.txt"), "rb").read() == data
    # level 0 (stored blocks) and level 9
    for lv in ("0", "9"):
        c = run([exe, *at, "-l", lv, "-c", src])
        assert run([exe, *at, "-d"], stdin=c) == data
        if checker:
            assert run([checker, "-d"], stdin=c) == data
    # bgzip -g: re-cut at the block boundaries of a .gzi (test_rebgzip); decoded bytes and block sizes must match
    t = os.path.join(tmp, "bgziptest.txt")
    open(t, "wb").write(open(os.path.join(GOLD, "bgziptest.txt.gz.plain"), "rb").read())
    want = open(os.path.join(GOLD, "bgziptest.txt.gz"), "rb").read()
    got = run([exe, *at, "-I", os.path.join(GOLD, "bgziptest.txt.gz.gzi"), "-c", "-g", t])
    assert [b[2] for b in refutil.split_blocks(got)] == [b[2] for b in refutil.split_blocks(want)]
    assert run([exe, "-d"], stdin=got) == open(t, "rb").read()
    # plain gzip input (not BGZF) is read like stock htslib reads it (bgzf.c:1165-1196)
    import gzip
    gz = gzip.compress(data[:700_000], 6) + gzip.compress(data[700_000:], 1)              # two members
    assert run([exe, *at, "-d"], stdin=gz) == data
