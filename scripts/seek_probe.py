"""GPU box: latency of random bgzf_seek + small bgzf_read through libhts_bgzf.so (and the reference library when built).
usage: seek_probe.py FILE.bgzf [nseek] [read_bytes]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import bgzf_capi
path = sys.argv[1]; nseek = int(sys.argv[2]) if len(sys.argv) > 2 else 1000; rd = int(sys.argv[3]) if len(sys.argv) > 3 else 100
raw = np.fromfile(path, dtype=np.uint8)
offs = []; pos = 0
while pos + 18 <= len(raw):
    bs = int(raw[pos + 16]) | (int(raw[pos + 17]) << 8); offs.append(pos); pos += bs + 1
rng = np.random.default_rng(7)
seeks = [(offs[i] << 16) | 10 for i in rng.integers(0, len(offs) - 2, nseek)]

def run(L, threads, name):
    fp = L.bgzf_open(path.encode(), b"r")
    if threads: L.bgzf_mt(fp, threads, 256)
    buf = C.create_string_buffer(max(rd, 128))
    lat = []
    for vo in seeks:
        t = time.perf_counter()
        ok = L.bgzf_seek(fp, vo, 0) == 0 and L.bgzf_read(fp, buf, rd) == rd
        lat.append(time.perf_counter() - t)
        assert ok
    L.bgzf_close(fp)
    lat = np.array(lat) * 1e6
    print(f"{name}: seek+read({rd} B) mean {lat.mean():.1f} us  median {np.median(lat):.1f}  p90 {np.percentile(lat, 90):.1f}  min {lat.min():.1f}")

run(bgzf_capi.load(), 4, "gpu ")
ref = os.path.join(ROOT, "oracle", "_ref", "libref_bgzf_ld.so")
if os.path.exists(ref):
    R = C.CDLL(ref); P = C.POINTER(bgzf_capi.BGZF)
    R.bgzf_open.restype = P; R.bgzf_open.argtypes = [C.c_char_p, C.c_char_p]; R.bgzf_close.argtypes = [P]
    R.bgzf_mt.argtypes = [P, C.c_int, C.c_int]
    R.bgzf_read.restype = C.c_ssize_t; R.bgzf_read.argtypes = [P, C.c_void_p, C.c_size_t]
    R.bgzf_seek.restype = C.c_int64; R.bgzf_seek.argtypes = [P, C.c_int64, C.c_int]
    run(R, 4, "ref4")
    run(R, 0, "ref0")
