"""include/hts_cram_gpu.h: cram_uncompress_block / cram_compress_block with the reference's own struct cram_block.
CPU: the struct / enum layout equals the reference's cram/cram_structs.h (checked by compiling against both headers).
GPU: a plain-C driver (tests/native/cram_blocks_c.c) pushes all 565 blocks of the reference's 34 CRAM v3.0 fixtures
through hg_cram_read_block -> hg_cram_write_block (byte-identical, CRC recomputed on the device) ->
cram_uncompress_block (array form, one by one, and from 8 threads at once = the coalescing path) and then through
the auto-tuning compressor and back."""
import json
import os
import struct
import subprocess

import pytest

from tests import refutil

ROOT = refutil.ROOT
NAT = os.path.join(ROOT, "tests", "native")
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference headers")
def test_struct_layout_equals_reference_headers(tmp_path, built):
    t = str(tmp_path)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/config.h"], check=False, capture_output=True)
    subprocess.run(["gcc", "-c", "-I", os.path.join(ROOT, "oracle", "_ref"), "-I", REF, os.path.join(NAT, "cram_layout_ref.c"), "-o", t + "/r.o"], check=True)
    subprocess.run(["gcc", "-c", "-I", os.path.join(ROOT, "include"), os.path.join(NAT, "cram_layout_ours.c"), "-o", t + "/o.o"], check=True)
    subprocess.run(["gcc", os.path.join(NAT, "cram_layout_main.c"), t + "/r.o", t + "/o.o", "-o", t + "/layout"], check=True)
    out = subprocess.run([t + "/layout"], capture_output=True, text=True)
    assert out.returncode == 0 and "layout identical" in out.stdout, out.stdout


def blocks():
    return json.load(open(os.path.join(refutil.ROOT, "tests", "golden", "cram_blocks.json")))


@pytest.fixture(scope="module")
def driver(built, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cramc") / "cram_blocks_c")
    lib = os.path.join(ROOT, "htslib_amd")
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(NAT, "cram_blocks_c.c"),
                    "-o", exe, "-L", lib, "-lhts_bgzf", "-lhtsgpu", "-lpthread", "-Wl,-rpath," + lib], check=True)
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["array", "single", "threads"])
def test_c_driver_decodes_every_reference_fixture_block(driver, engine, tmp_path, mode):
    bl = blocks()
    disk = b"".join(bytes.fromhex(b["hdr_hex"]) + bytes.fromhex(b["data_hex"]) + struct.pack("<I", b["crc32"]) for b in bl)
    src, out = str(tmp_path / "blocks.bin"), str(tmp_path / "out.bin")
    open(src, "wb").write(disk)
    r = subprocess.run([driver, src, out, mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-500:]
    assert f"blocks {len(bl)} decoded_ok {len(bl)}" in r.stdout, r.stdout
    assert open(out + ".rewrite", "rb").read() == disk                      # hg_cram_write_block reproduces the file bytes
    raw = open(out, "rb").read()
    pos = 0
    pinned = 0
    for b in bl:
        rc, n = struct.unpack_from("<ii", raw, pos); pos += 8
        assert rc == 0 and n == b["usize"], (b["source"], b["method"], b["content_id"])
        got = raw[pos:pos + n]; pos += n
        if b["expected_hex"] is not None:                                   # plaintext known without any code of ours
            assert got == bytes.fromhex(b["expected_hex"]), (b["source"], b["method"], b["content_id"])
            pinned += 1
    assert pos == len(raw) and pinned >= 500


@pytest.mark.gpu
def test_c_driver_reports_corrupt_blocks_like_the_reference(driver, engine, tmp_path):
    """A flipped payload byte = "Block CRC32 failure" -> -1 for that block only (cram_io.c:1585-1592)."""
    bl = [b for b in blocks() if b["method"] in (1, 4) and len(b["data_hex"]) >= 16][:40]
    parts = []
    for i, b in enumerate(bl):
        data = bytearray(bytes.fromhex(b["data_hex"]))
        if i % 5 == 0: data[len(data) // 2] ^= 0x40
        parts.append(bytes.fromhex(b["hdr_hex"]) + bytes(data) + struct.pack("<I", b["crc32"]))
    src, out = str(tmp_path / "blocks.bin"), str(tmp_path / "out.bin")
    open(src, "wb").write(b"".join(parts))
    r = subprocess.run([driver, src, out, "threads"], capture_output=True, text=True, timeout=600, env=dict(os.environ, CRAMC_ALLOW_BAD_CRC="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count("Block CRC32 failure") == 8
    raw = open(out, "rb").read()
    pos = 0
    for i, b in enumerate(bl):
        rc, n = struct.unpack_from("<ii", raw, pos); pos += 8 + n
        assert rc == (-1 if i % 5 == 0 else 0) and n == (0 if i % 5 == 0 else b["usize"])


@pytest.fixture(scope="module")
def slice_driver(built, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cramsl") / "cram_slice_c")
    lib = os.path.join(ROOT, "htslib_amd")
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(NAT, "cram_slice_c.c"),
                    "-o", exe, "-L", lib, "-lhts_bgzf", "-lhtsgpu", "-lpthread", "-Wl,-rpath," + lib], check=True)
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("level,major,use_fqz,use_arith", [(5, 3, 0, 0), (7, 3, 1, 1), (1, 3, 0, 0), (5, 2, 0, 0), (5, 2, 2, 0)])
def test_c_driver_compress_slice_in_one_batch(slice_driver, engine, level, major, use_fqz, use_arith):
    """hg_cram_compress_slice_fqz = cram_compress_slice (cram/cram_encode.c:803-988) for one slice: every block decodes back to its bytes with
    cram_uncompress_block, every chosen method belongs to the set hg_cram_slice_plan offers that series (or methodF of the final sweep), the
    metrics objects carry over five slices (trial phase, then the learnt method); with use_fqz the quality block may come out as method 7"""
    r = subprocess.run([slice_driver, str(level), str(major), str(use_fqz), str(use_arith), "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-800:]
    assert "undecodable 0" in r.stdout and "methods_offered_ok" in r.stdout
    rows = [l.split() for l in r.stdout.splitlines() if l.startswith("slice")]
    assert len(rows) == 5 * 8
    methods = {(int(x[3]), int(x[5])) for x in rows}                     # (ds, on-disk method)
    qs = {m for ds, m in methods if ds == 12}
    assert all(int(x[9]) <= int(x[7]) for x in rows)                      # nothing grows (RAW is kept when nothing beats it)
    if major >= 3:
        assert 8 in {m for ds, m in methods if ds == 11}                  # read names: the tokeniser
        assert qs <= ({5, 7} if use_fqz & 1 else {5}) | ({6} if use_arith else set()) | {1, 4}
    else:
        assert not ({5, 6, 7, 8} & {m for _, m in methods})               # CRAM 2.x: gzip / rANS 4x8 only (+ bzip2 / lzma when asked for and installed)
        if use_fqz & 2:
            import ctypes
            have = []
            for name, mid in (("libbz2.so.1.0", 2), ("liblzma.so.5", 3)):
                try: ctypes.CDLL(name); have.append(mid)
                except OSError: pass
            print("bzip2 / lzma available:", have, "chosen somewhere:", sorted({m for _, m in methods} & {2, 3}))
            assert {m for _, m in methods} & {2, 3} <= set(have)


@pytest.mark.gpu
def test_reference_named_entry_points_on_a_real_cram_fd(engine, tmp_path):
    """cram_compress_block / cram_compress_block2 / cram_write_block / cram_read_block / cram_uncompress_block by their reference names on the
    reference's REAL struct cram_fd and cram_slice: tests/native/cram_fd_driver.c is compiled against /root/reference's headers (oracle/Makefile,
    where the reference is present) and linked to our library; the offsets it relies on are asserted by test_struct_layout_equals_reference_headers."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cram_fd_driver")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/cram_fd_driver not built (needs the reference headers at build time)")
    r = subprocess.run([exe, str(tmp_path / "blocks.cram")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cram_fd entry points ok" in r.stdout, r.stderr[-1500:] + r.stdout[-500:]
    meth = [int(x) for x in r.stdout.split()[1:4]]
    assert meth[1] in (5, 7) and meth[2] == 8, meth                      # qualities: rANS Nx16 or fqzcomp; names: the tokeniser
