"""Probe of the two-phase range-coder encoder and the decoder behind it: the cases of tests/test_arith.py's two-phase test, one report line per failure (GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from htslib_amd import _native as nat
from tests import refutil

eng = nat.Engine(0)
orc = refutil.ArithOracle()
rng = np.random.default_rng(77)
datas, flags = [], []
for n in (8192, 8193, 8255, 8256, 8257, 20_000, 300_000):
    for m in (1, 2, 40, 64, 65, 130, 256):
        p = rng.dirichlet(np.full(m, 0.3)) if m > 1 else np.ones(1)
        d = bytes(rng.choice(m, n, p=p).astype(np.uint8))
        if m == 256: d = bytes([255]) + d[1:]
        for fl in (0, 1) if n != 300_000 else (0, 1, 9, 8, 64, 65, 128, 129, 193):
            datas.append(d); flags.append(fl)
t = time.perf_counter()
enc = eng.arith_encode_host(datas, flags)
print("encode call %.1f ms" % ((time.perf_counter() - t) * 1e3))
outs, st = eng.cram_uncompress_blocks([(6, e, len(d)) for d, e in zip(datas, enc)])
for d, fl, e, o, s in zip(datas, flags, enc, outs, st):
    ref = orc.encode(d, fl)
    rc, back = orc.decode(e, len(d), -1)
    if e != ref or s != 0 or o != d:
        print("n=%d m=%d flags=%d: encoder %s, gpu decoder status %d %s, oracle decodes it: %s" % (len(d), max(d) + 1, fl, "ok" if e == ref else "DIFFERENT", s, "ok" if o == d else "WRONG", rc == 0 and back == d))
print("done", len(datas))
