"""GPU box: throughput probe of the CRAM record decoder (hg_cram_decode_records_host, PCIe included) on a batch of slices -- the three
50-record multi-reference slices of the reference's range.cram and the tlen fixtures, replicated.  usage: bench_cram_records.py [slices]
Both mappings (one wavefront per slice / one slice per lane) are timed, and the same source compiled for one CPU core beside them."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from htslib_amd import _native as nat
from tests import test_cram_records as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
SYNTH = int(sys.argv[2]) if len(sys.argv) > 2 else 0            # records per synthetic slice (htslib_amd/synth_cram.py); 0 = the range.cram fixtures
eng = nat.Engine(0)
if SYNTH:
    from htslib_amd import synth_cram as cram_synth
    rng = np.random.default_rng(5)
    base = [cram_synth.make_slice(rng, SYNTH, 150) for _ in range(4)]
    NREF = 1
else:
    base = [s for fname, major, nref, s in T.load_slices() if fname == "test/range.cram"]
    NREF = 7
slices = [base[i % len(base)] for i in range(N)]
nrec = sum(s["nrec"] for s in slices); nbases = sum(len(e[9]) for s in slices for e in s["expect"] if e[9] != "*")
bound, dec = T._gpu_calls(eng)
so = "/tmp/libcram_records_host.so"
subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, "tests/native/cram_records_host.cpp"], check=True)
L = C.CDLL(so)
L.hgr_host_records_bound.argtypes = [C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
L.hgr_host_decode_records.argtypes = [C.c_size_t, C.c_void_p, C.c_int, C.c_int] + [C.c_size_t] * 5 + [C.c_void_p] * 3

def timed(label, call_bound, call_dec, reps=2):
    ts = []
    def wrapped(*a):
        t = time.perf_counter(); r = call_dec(*a); ts.append(time.perf_counter() - t); return r
    for _ in range(reps):
        st, got = T.decode(call_bound, wrapped, slices, 3, NREF)
    assert (st == 0).all()
    if SYNTH: T._check_truth(slices[:2], got[:2])
    else: T.check_against_twin("bench", got[0], slices[0]["expect"]); T.check_against_twin("bench", got[-1], slices[-1]["expect"])
    print("%-34s %8.2f M records/s  %7.1f M bases/s  (%d slices, %d records, best of %d: %.1f ms in the call)"
          % (label, nrec / min(ts) / 1e6, nbases / min(ts) / 1e6, N, nrec, reps, min(ts) * 1e3), flush=True)

for mode in ("wave", "lane"):
    os.environ["HG_CRAM_RECORDS_MODE"] = mode
    timed("GPU, mapping = %s" % mode, bound, dec)
timed("same source, one CPU core", L.hgr_host_records_bound, L.hgr_host_decode_records, reps=1)
