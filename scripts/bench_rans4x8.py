"""GPU box: throughput probe of the CRAM 3.0 rANS 4x8 codec through the host entry points (PCIe included) -- orders 0 and 1.
usage: bench_rans4x8.py [streams] [bytes per stream]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from htslib_amd import _native as nat
from tests.test_rans4x8 import synth_series
eng = nat.Engine(0)
rng = np.random.default_rng(1)
NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
QLEN = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
base = [synth_series(rng, "qual41", QLEN) for _ in range(16)]
quals = [base[i % 16] for i in range(NQ)]
qb = NQ * QLEN
def timed(label, fn, reps=3):
    fn(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
    print("%-28s %8.2f GB/s (host API, %d streams x %d B = %d MB, best of %d: %.1f ms)" % (label, qb / min(ts) / 1e9, NQ, QLEN, qb >> 20, reps, min(ts) * 1e3), flush=True)
    return r
for order in (0, 1):
    enc = timed("rans4x8 encode order %d" % order, lambda: eng.rans4x8_encode_host(quals, [order] * NQ))
    out = timed("rans4x8 decode order %d" % order, lambda: eng.cram_uncompress_blocks([(4, e, QLEN) for e in enc]))
    assert out[0][0] == quals[0] and (out[1] == 0).all()
