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
