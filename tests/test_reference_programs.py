"""The drop-in proof on the MI355X: the reference's OWN test/test_bgzf.c and bgzip.c -- compiled unmodified from
/root/reference against the reference's headers by oracle/Makefile (target `dropin`) -- linked to OUR front-end
(htslib_amd/libhts_bgzf.so -> libhtsgpu.so) and run the way the reference's harness runs them.
Two link flavours: the bundled hFILE provider (hfile_min.cpp) and the reference's real hfile.c (`_refhfile`)."""
import os

import pytest

from tests import dropin_cases, refutil

pytestmark = pytest.mark.gpu
D = refutil.REF_DIR


def need(name):
    p = os.path.join(D, name)
    if not os.path.exists(p):
        pytest.skip(f"{p} was not built (oracle/Makefile dropin needs /root/reference at build time)")
    return p


@pytest.mark.parametrize("flavour", ["", "_refhfile"])
def test_reference_test_bgzf_passes_unmodified(built, engine, tmp_path, flavour):
    dropin_cases.reference_test_bgzf(need("test_bgzf_gpu" + flavour), str(tmp_path))


@pytest.mark.parametrize("flavour,threads", [("", 0), ("", 4), ("_refhfile", 4)])
def test_reference_bgzip_on_our_library(built, engine, tmp_path, flavour, threads):
    checker = os.path.join(D, "ref_bgzip_ld") if refutil.have_ref() else None
    dropin_cases.reference_bgzip(need("bgzip_gpu" + flavour), str(tmp_path), threads, checker)


def test_reference_bgzip_over_several_device_contexts_is_byte_identical(built, engine, tmp_path):
    """SURVEY 8e inside the library: HTS_GPU_DEVICES names the devices of a handle; batches rotate over them and come back in submission order.  "0,0" = two
    contexts on the one GPU of the test box (an entry may repeat a device): the reference's bgzip on OUR library must write the SAME compressed bytes and
    .gzi as with one device, and read them back -- the real libhts_bgzf.so / libhtsgpu.so, not the CPU test double."""
    import numpy as np, subprocess
    exe = need("bgzip_gpu")
    from htslib_amd import synth
    data, _, _ = synth.bam_stream(96 << 20, 0x5EED0001, 0, True)           # enough for many batches (writer jobs of up to 768 blocks, 8 MiB reader windows)
    src = tmp_path / "in.bam.raw"; src.write_bytes(data)
    outs = {}
    for tag, devs in (("one", "0"), ("two", "0,0"), ("three", "0,0,0")):
        env = dict(os.environ, HTS_GPU_DEVICES=devs)
        comp, idx = tmp_path / (tag + ".gz"), tmp_path / (tag + ".gzi")
        with open(src, "rb") as fi, open(comp, "wb") as fo:
            r = subprocess.run([exe, "-@4", "-i", "-I", str(idx), "-c"], stdin=fi, stdout=fo, stderr=subprocess.PIPE, env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
        with open(comp, "rb") as fi:
            r = subprocess.run([exe, "-@4", "-d", "-c"], stdin=fi, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        assert r.returncode == 0 and r.stdout == data, (tag, r.stderr.decode(errors="replace")[-2000:])
        outs[tag] = (comp.read_bytes(), idx.read_bytes())
    assert outs["one"] == outs["two"] == outs["three"]
    if refutil.have_ref():                                                  # and stock htslib reads it
        r = subprocess.run([os.path.join(D, "ref_bgzip_ld"), "-d", "-c", str(tmp_path / "two.gz")], stdout=subprocess.PIPE, timeout=600)
        assert r.returncode == 0 and r.stdout == data
