"""ASan fuzz of the CRAM record decoder sources on the CPU (chain + data-parallel passes): mutated slices must come back with a status, never touch
memory out of bounds (on the device a stray read is a fault).  usage: LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python scripts/fuzz_cram_fast.py [iterations]"""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_cram_records as T
from tests.test_cram_records_fast import fast_call
from htslib_amd import synth_cram

_vp = C.c_void_p
so = "/tmp/libcram_records_host_asan.so"
subprocess.run(["g++", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so,
                os.path.join(T.ROOT, "tests", "native", "cram_records_host.cpp")], check=True)
L = C.CDLL(so)
L.hgr_host_records_bound.argtypes = [C.c_size_t, _vp, C.c_int, _vp, _vp, _vp, _vp]
L.hgr_host_decode_records.argtypes = [C.c_size_t, _vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp]
L.hgr_host_decode_records_fast.argtypes = L.hgr_host_decode_records.argtypes + [_vp]
fc = fast_call(L)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
base = [synth_cram.make_slice(rng, 60, 70, tags=True), synth_cram.make_slice(rng, 45, 50, unmapped_every=4), synth_cram.make_slice(rng, 30, 64, detached_every=2, tags=True)]
base += [s for _, _, _, s in T.load_slices()]

def mutate(b):
    b = bytearray(b)
    if not b: return bytes(b)
    k = int(rng.integers(0, 4))
    if k == 0:
        for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif k == 1: b = b[:int(rng.integers(0, len(b)))]
    elif k == 2: b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    else:
        i = int(rng.integers(0, len(b))); b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8))
    return bytes(b)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
took = ok = bad = 0
for it in range(n):
    s = dict(base[int(rng.integers(0, len(base)))])
    what = int(rng.integers(0, 6))
    if what == 0: s["comp_hdr"] = mutate(s["comp_hdr"])
    elif what == 1: s["slice_hdr"] = mutate(s["slice_hdr"])
    elif what <= 4 and s["blocks"]:
        j = int(rng.integers(0, len(s["blocks"]))); bl = list(s["blocks"]); bl[j] = (bl[j][0], mutate(bl[j][1])); s["blocks"] = bl
    else: s["refs"] = [(t, a, b[:len(b) // 2], ln) for t, a, b, ln in s.get("refs", [])]
    res = []
    for call in (L.hgr_host_decode_records, fc):
        try:
            st, got = T.decode(L.hgr_host_records_bound, call, [s], 3, 7)
            res.append((int(st[0]), got if st[0] == 0 else None))
        except AssertionError: res.append(("refused",))
        except (ValueError, IndexError, UnicodeDecodeError, T.struct_error): res.append(("unrenderable",))
    assert res[0] == res[1], (it, what, res[0][0], res[1][0])
    if res[1][0] == 0: ok += 1; took += int(fc.path[0])
    else: bad += 1
print("iterations %d: decoded %d (%d by the data-parallel passes), rejected %d, chain == passes every time, no ASan report" % (n, ok, took, bad))
