"""GPU box: bgzf_write loop through libhts_bgzf.so (level 6, bgzf_mt so that the writer batches) on plain BAM bytes; prints GB/s."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import bgzf_capi
from htslib_amd import synth
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2
plain, _, _ = synth.bam_stream(int(gib * (1 << 30)), 0x5EED0001, 0, True)
host = np.frombuffer(plain, dtype=np.uint8)
L = bgzf_capi.load()
chunk = 8 << 20
for rep in range(2):
    t0 = time.perf_counter()
    fp = L.bgzf_open(b"/dev/shm/write_timeline.bam", b"w")
    L.bgzf_mt(fp, 4, 256)
    pos = 0
    while pos < len(plain):
        n = min(chunk, len(plain) - pos)
        assert L.bgzf_write(fp, C.cast(host.ctypes.data + pos, C.c_char_p), n) == n
        pos += n
    t1 = time.perf_counter()
    L.bgzf_close(fp)
    t2 = time.perf_counter()
    print(f"rep {rep}: write loop {t1 - t0:.3f} s, close {t2 - t1:.3f} s, overall {len(plain) / (t2 - t0) / 1e9:.2f} GB/s")
os.unlink("/dev/shm/write_timeline.bam")
