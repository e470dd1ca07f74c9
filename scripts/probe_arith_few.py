"""A call with few range-coder streams is latency-bound: the encoder takes its two-phase form from 16 KiB (HG_ARITH_2P_FEW / HG_ARITH_2P_MIN_FEW).  One and
eight streams of 100 000 symbols against the oracle, with the call's wall time; HG_ARITH_2P=0 in the environment gives the one-pass figures (GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from htslib_amd import _native as nat
from tests import refutil

eng = nat.Engine(0)
orc = refutil.ArithOracle()
rng = np.random.default_rng(11)
for m in (40, 256):
    p = rng.dirichlet(np.full(m, 0.5))
    for k in (1, 8):
        ds = [bytes(rng.choice(m, 100_000, p=p).astype(np.uint8)) for _ in range(k)]
        for fl in (0, 1, 65):
            eng.arith_encode_host(ds, [fl] * k)
            t = time.perf_counter(); e = eng.arith_encode_host(ds, [fl] * k); dt = time.perf_counter() - t
            ok = all(x == orc.encode(d, fl) for x, d in zip(e, ds))
            print("alphabet %3d, %d stream(s) x 100 000, flags %2d: %6.1f ms %s" % (m, k, fl, dt * 1e3, "ok" if ok else "DIFFERENT"), flush=True)
