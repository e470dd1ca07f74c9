"""CPU check of the BGZF front-end's HOST logic (state machine, I/O / output threads, seek, index, EOF handling):
the reference's own test/test_bgzf.c and bgzip.c, compiled unmodified, run against  bgzf_front.cpp + hfile_min.cpp  on
the zlib TEST DOUBLE of the engine (tests/native/fake_engine.c).  The product library is not involved: it links the
gfx950 engine and refuses to open compressed streams without a GPU (tests/test_cabi.py::test_no_gpu_fails_loudly).
The same scenarios run against the real library on the MI355X in tests/test_reference_programs.py."""
import os
import subprocess

import pytest

from tests import dropin_cases, refutil

ROOT = refutil.ROOT
OUT = os.path.join(ROOT, "build", "hostlogic")
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference sources to compile test_bgzf.c / bgzip.c")


@pytest.fixture(scope="module", params=["", "thread", "address,undefined"])
def progs(request):
    san = request.param
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "native", "build_hostlogic.sh"), *([san] if san else [])],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    suf = f"_{san}" if san else ""
    return os.path.join(OUT, "test_bgzf_fake" + suf), os.path.join(OUT, "bgzip_fake" + suf)


def test_reference_test_bgzf_passes_on_the_front_end(progs, tmp_path):
    dropin_cases.reference_test_bgzf(progs[0], str(tmp_path))


@pytest.mark.parametrize("threads", [0, 4])
def test_reference_bgzip_scenarios(progs, tmp_path, threads):
    checker = os.path.join(refutil.REF_DIR, "ref_bgzip") if refutil.have_ref() else None
    dropin_cases.reference_bgzip(progs[1], str(tmp_path), threads, checker)


def test_large_reads_go_through_the_copy_helpers(progs, tmp_path):
    """bgzf_read with an 8 MiB buffer: copies of 2 MiB and more are shared with helper threads (race detector on in the
    `thread` build); the bytes must be the file's."""
    import numpy as np
    suf = "_thread" if progs[0].endswith("_thread") else "_address,undefined" if progs[0].endswith("_address,undefined") else ""
    exe = str(tmp_path / "bigread")
    r = subprocess.run(["gcc", "-O1", "-g"] + (["-fsanitize=" + suf[1:]] if suf else []) +
                       ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "bigread.c"), "-o", exe,
                        "-L", OUT, "-lhts_bgzf_fake" + suf, "-Wl,-rpath," + OUT, "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    data = (np.random.default_rng(3).integers(0, 64, 30_000_000, dtype=np.uint8) + 32).tobytes()
    plain, gz = tmp_path / "big.txt", tmp_path / "big.gz"
    plain.write_bytes(data)
    with open(gz, "wb") as f:
        assert subprocess.run([progs[1], "-@4", "-c", str(plain)], stdout=f).returncode == 0
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe, str(gz)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert out.stdout == data


def test_bulk_writes_take_whole_blocks_from_the_callers_buffer(progs, tmp_path):
    """bgzf_write after bgzf_mt() with calls from 3 bytes to 8 MiB: whole blocks skip fp->uncompressed_block (and spans of megabytes use the copy helpers;
    race detector on in the `thread` build).  Same cuts as a writer that only ever sees 1000-byte calls: identical file and .gzi; the file decompresses."""
    import numpy as np
    suf = "_thread" if progs[0].endswith("_thread") else ""
    exe = str(tmp_path / "bigwrite")
    r = subprocess.run(["gcc", "-O1", "-g"] + (["-fsanitize=thread"] if suf else []) +
                       ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "bigwrite.c"), "-o", exe,
                        "-L", OUT, "-lhts_bgzf_fake" + suf, "-Wl,-rpath," + OUT, "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    data = (np.random.default_rng(4).integers(0, 64, 40_000_123, dtype=np.uint8) + 32).tobytes()
    plain = tmp_path / "big.txt"; plain.write_bytes(data)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    outs = []
    for mode in ("0", "1"):
        gz = tmp_path / f"w{mode}.gz"
        r = subprocess.run([exe, str(plain), str(gz), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, (r.returncode, r.stderr.decode()[-2000:])
        outs.append((gz.read_bytes(), (tmp_path / f"w{mode}.gz.gzi").read_bytes()))
    assert outs[0] == outs[1]
    r = subprocess.run([progs[1], "-@4", "-d", "-c", str(tmp_path / "w0.gz")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0 and r.stdout == data, r.stderr.decode()[-1500:]


def test_a_handle_spreads_its_batches_over_several_devices(progs, tmp_path):
    """HTS_GPU_DEVICES=0-3: one context per device, two pipes each, windows rotate over the devices and come back in submission order (static
    split, no collective).  The reference's bgzip on the front-end + the test double with four "devices": the compressed file is byte-identical to the
    one-device run, decompression gives the input back, and every device took batches in both directions."""
    import numpy as np
    data = (np.random.default_rng(9).integers(0, 40, 40_000_000, dtype=np.uint8) + 48).tobytes()
    plain = tmp_path / "in.txt"; plain.write_bytes(data)
    outs = {}
    for devs in ("0", "0-3"):
        env = dict(os.environ, HTS_GPU_DEVICES=devs, FAKE_ENGINE_REPORT="1", TSAN_OPTIONS="halt_on_error=1")
        gz = tmp_path / f"out_{len(devs)}.gz"
        with open(gz, "wb") as f:
            r = subprocess.run([progs[1], "-@4", "-c", str(plain)], stdout=f, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        rep_c = [l for l in r.stderr.decode().splitlines() if "jobs per device" in l][-1]
        r = subprocess.run([progs[1], "-@4", "-d", "-c", str(gz)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0 and r.stdout == data, r.stderr.decode()[-1500:]
        rep_d = [l for l in r.stderr.decode().splitlines() if "jobs per device" in l][-1]
        outs[devs] = (gz.read_bytes(), rep_c, rep_d)
    assert outs["0"][0] == outs["0-3"][0]                                 # same bytes whichever devices did the work
    for rep in outs["0-3"][1:]:
        used = {int(x.split("=")[0]): int(x.split("=")[1]) for x in rep.split(":")[1].split()}
        assert set(used) == {0, 1, 2, 3} and min(used.values()) >= 1, rep
    assert set(int(x.split("=")[0]) for x in outs["0"][1].split(":")[1].split()) == {0}


# ---- the front-end inside the reference's whole libhts (build_hostlogic.sh: libhts_fake.so = reference objects minus bgzf.o + bgzf_front.cpp on the test
# double): sam.c's bam_read1 / bam_write1, vcf.c's BCF reader / writer and the on-the-fly indexes over our bgzf_* host logic.  GPU twin: tests/test_libhts_gpu.py
@pytest.fixture(scope="module")
def libhts_fake(tmp_path_factory):
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "native", "build_hostlogic.sh")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    v, i = os.path.join(OUT, "test_view_fake"), os.path.join(OUT, "test_index_fake")
    if not (os.path.exists(v) and os.path.exists(i)): pytest.skip("oracle/_ref/hts_obj not built")
    from tests import test_libhts_gpu as L
    if not os.path.exists(L.VIEW_REF): pytest.skip("oracle/_ref/ref_view not built")
    return L, v, i, L.unpack_fixtures(str(tmp_path_factory.mktemp("viewfix")))


@pytest.mark.parametrize("threads", [0, 4])
def test_reference_test_index_scenarios_on_the_front_end_inside_libhts(libhts_fake, threads):
    L, v, i, d = libhts_fake
    L.index_scenarios(d, threads, v, i)


def test_reference_test_view_bam_scenarios_on_the_front_end_inside_libhts(libhts_fake):
    L, v, i, d = libhts_fake
    L.bam_both_directions(d, 4, v)
