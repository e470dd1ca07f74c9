"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/htsgpu.h declares, and its host-side framing scan agrees with the oracle.  No compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests import refutil
from htslib_amd import synth

ROOT = refutil.ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "htsgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    lib = C.CDLL(os.path.join(ROOT, "htslib_amd", "libhtsgpu.so"))
    syms = declared_symbols()
    assert len(syms) >= 9
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/htsgpu.h but not exported"
    from htslib_amd import _native
    assert set(_native.EXPORTS) == set(syms)


def test_version_and_strerror(built):
    from htslib_amd import _native as nat
    assert b"gfx950" in nat.lib.hg_version()
    assert nat.lib.hg_strerror(0) == b"ok"
    assert b"CPU" not in nat.lib.hg_strerror(-2)


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from htslib_amd import _native as nat
    with pytest.raises(nat.HgError) as e:
        nat.Engine(0)
    assert e.value.code == -2           # HG_ENODEV: there is no CPU fallback


def test_scan_matches_oracle_on_fixtures(built, oracle):
    from htslib_amd import _native as nat
    for name, comp, plain in refutil.golden_cases():
        desc, total = nat.bgzf_scan(comp)
        ref = refutil.split_blocks(comp)
        assert total == len(plain)
        assert [(int(d["coff"]), int(d["clen"]), int(d["ulen"])) for d in desc] == ref
        assert np.array_equal(desc["uoff"], np.concatenate([[0], np.cumsum(desc["ulen"])[:-1]]).astype(np.uint64))


def test_scan_rejects_bad_framing(built):
    from htslib_amd import _native as nat
    _, bg = synth.bam_bgzf(200_000)
    nat.bgzf_scan(bg)
    for bad in (bg[:-5], b"\x1f\x8b\x08\x00" + bg[4:], bg[:100], bg + b"junk"):
        with pytest.raises(nat.HgError) as e:
            nat.bgzf_scan(bad)
        assert e.value.code == -4       # HG_EFORMAT
    d, t = nat.bgzf_scan(b"")
    assert len(d) == 0 and t == 0
