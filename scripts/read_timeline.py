"""GPU box: bgzf_read loop through libhts_bgzf.so on a /dev/shm file; prints the cumulative time at every GiB (fixed costs vs steady state)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tests import bgzf_capi
from htslib_amd import synth
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
comp = bench.prepare(0x5EED0001, int(gib * (1 << 30)), 6, max(1, (os.cpu_count() or 2) - 4), None)
path = "/dev/shm/read_timeline.bam"
open(path, "wb").write(comp + synth.BGZF_EOF)
L = bgzf_capi.load()
chunk = 8 << 20
buf = C.create_string_buffer(chunk)
for rep in range(2):
    t0 = time.perf_counter()
    fp = L.bgzf_open(path.encode(), b"r")
    t_open = time.perf_counter() - t0
    tot = 0; marks = []; nxt = 1 << 30; first = None
    while True:
        n = L.bgzf_read(fp, buf, chunk)
        if first is None: first = time.perf_counter() - t0
        if n <= 0: break
        tot += n
        if tot >= nxt: marks.append(round(time.perf_counter() - t0, 4)); nxt += 1 << 30
    t_all = time.perf_counter() - t0
    L.bgzf_close(fp)
    t_close = time.perf_counter() - t0
    steady = (len(marks) - 1) * (1 << 30) / (marks[-1] - marks[0]) / 1e9 if len(marks) > 1 else None
    print(f"rep {rep}: open {t_open*1e3:.1f} ms, first read returns at {first*1e3:.1f} ms, GiB marks {marks}, all {t_all:.3f} s, close at {t_close:.3f}; overall {tot/t_all/1e9:.1f} GB/s, steady {steady and round(steady,1)} GB/s")
os.unlink(path)
