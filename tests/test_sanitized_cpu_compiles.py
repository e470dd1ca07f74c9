"""The device sources that also compile for the CPU (the CRAM record decoder -- chain and data-parallel passes -- and the record encoder: cram_records_core.h,
cram_records_fast.h, cram_encode_core.h through tests/native/cram_records_host.cpp) run their CPU tests once more under AddressSanitizer + UBSan: the
reference's 34 fixtures, the damaged-input set and an encode -> decode round trip must not touch a byte outside their buffers or hit undefined behaviour.
The tests run in a child pytest with libasan preloaded (the fixtures add HG_TEST_HOSTLIB_FLAGS to their compile lines)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_compiles_of_the_record_layer_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE).stdout.decode().strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan.so beside this gcc")
    env = dict(os.environ, HG_TEST_HOSTLIB_FLAGS="-g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    picks = ["tests/test_cram_records.py::test_damaged_inputs_are_rejected_or_decoded_never_fatal", "tests/test_cram_records.py::test_decoder_source_on_the_cpu_matches_the_sam_twins",
             "tests/test_cram_encode.py::test_synthetic_slices_survive_encode_and_decode"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + picks, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "3 passed" in out, out[-4000:]
