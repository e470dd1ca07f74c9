"""GPU box: what bounds bgzf_write?  The same bgzf_write loop (level 6, bgzf_mt, 8 MiB calls) into a /dev/shm file and into /dev/null:
the difference is the file system's share (one output thread hwrite()s the compressed stream).  usage: write_probe.py [GiB]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import bgzf_capi
from htslib_amd import synth
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1
plain, _, _ = synth.bam_stream(int(gib * (1 << 30)), 0x5EED0001, 0, True)
host = np.frombuffer(plain, dtype=np.uint8)
L = bgzf_capi.load()
chunk = 8 << 20
for path in (b"/dev/shm/write_probe.bam", b"/dev/null", b"/dev/shm/write_probe.bam", b"/dev/null"):
    t0 = time.perf_counter()
    fp = L.bgzf_open(path, b"w")
    L.bgzf_mt(fp, 4, 256)
    pos = 0
    while pos < len(plain):
        n = min(chunk, len(plain) - pos)
        assert L.bgzf_write(fp, C.cast(host.ctypes.data + pos, C.c_char_p), n) == n
        pos += n
    L.bgzf_close(fp)
    t2 = time.perf_counter()
    print(f"{path.decode():28s} {len(plain) / (t2 - t0) / 1e9:6.2f} GB/s")
if os.path.exists("/dev/shm/write_probe.bam"): os.unlink("/dev/shm/write_probe.bam")
