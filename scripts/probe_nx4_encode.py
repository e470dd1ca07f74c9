"""4-way rANS Nx16 encode through the host API: call times for big and small streams, order 0 / 1 (HG_NX4_SCALAR=0 in the environment: the lane-group coder).  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from htslib_amd import _native as nat
eng = nat.Engine(0)
rng = np.random.default_rng(2)
def markov(n, levels, stay):
    ch = rng.random(n) >= stay; idx = np.maximum.accumulate(np.where(ch, np.arange(n), 0)); return rng.integers(0, levels, n).astype(np.uint8)[idx]
big = markov(1_500_000, 40, 0.7); small = rng.integers(0, 200, 40_000).astype(np.uint8)
eng.ransnx16_encode_host([bytes(1000)], [0])
for name, d, cnt in (("1.5 MB x 256", big, 256), ("40 KB x 2048", small, 2048)):
    ds = [bytes(np.roll(d, 13 * i)) for i in range(cnt)]
    for fl in (0, 1):
        ts = []
        for _ in range(3):
            t = time.perf_counter(); enc = eng.ransnx16_encode_host(ds, [fl] * cnt); ts.append(time.perf_counter() - t)
        tot = sum(map(len, ds))
        print("%-14s order %d: %.1f ms per call = %.2f GB/s, ratio %.3f" % (name, fl, min(ts) * 1e3, tot / min(ts) / 1e9, sum(map(len, enc)) / tot), flush=True)
