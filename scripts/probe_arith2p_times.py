"""Per-kernel times of the two-phase range-coder encoder (HG_ARITH_2P_TIMES=1) on streams of known shape: one call per shape, 64 equal streams each.  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
os.environ["HG_ARITH_2P_TIMES"] = "1"; os.environ["HG_ARITH_2P_MIN"] = "1"
from htslib_amd import _native as nat
eng = nat.Engine(0)
rng = np.random.default_rng(5)
def markov(n, levels, stay):
    ch = rng.random(n) >= stay; idx = np.maximum.accumulate(np.where(ch, np.arange(n), 0)); return rng.integers(0, levels, n).astype(np.uint8)[idx]
shapes = {"40 levels iid 100k": rng.integers(0, 40, 100_000).astype(np.uint8), "4 levels markov 100k": markov(100_000, 4, 0.9), "256 flat 100k": rng.integers(0, 256, 100_000).astype(np.uint8),
          "40 levels markov 1.5M": markov(1_500_000, 40, 0.7), "one symbol 100k": np.zeros(100_000, np.uint8)}
eng.arith_encode_host([bytes(1000)], [1])
for name, d in shapes.items():
    for fl in (0, 1, 64, 65):
        ds = [bytes(np.roll(d, 17 * i)) for i in range(16 if len(d) > 200_000 else 64)]
        print("== %s, flags %d, %d streams" % (name, fl, len(ds)), file=sys.stderr, flush=True)
        t = time.perf_counter(); enc = eng.arith_encode_host(ds, [fl] * len(ds)); dt = time.perf_counter() - t
        print("   call %.1f ms, ratio %.3f" % (dt * 1e3, sum(map(len, enc)) / sum(map(len, ds))), file=sys.stderr, flush=True)
