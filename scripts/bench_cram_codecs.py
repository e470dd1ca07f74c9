"""Throughput probe for the CRAM 3.1 codecs that have no bench.py line of their own (range coder, tok3, Nx16
transforms).  Uses the synchronous host entry points, so wall time includes PCIe copies; run it under
`rocprofv3 --kernel-trace --stats` for kernel-only durations.  Inputs are produced by the GPU encoders
(the oracle is not involved)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401  (initialises the HIP runtime the same way the tests do)
from htslib_amd import _native as nat
from tests.test_rans4x8 import synth_series
from tests.test_tok3 import illumina_names

eng = nat.Engine(0)
rng = np.random.default_rng(1)
NQ, QLEN = 1024, 300_000
base = [synth_series(rng, "qual41", QLEN) for _ in range(16)]
quals = [base[i % 16] for i in range(NQ)]
names = [illumina_names(np.random.default_rng(i), 10_000) for i in range(8)]
NB = 512
nameblocks = [names[i % 8] for i in range(NB)]


def timed(label, fn, nbytes, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
    print("%-34s %8.2f GB/s (host API, %d MB, best of %d: %.1f ms)" % (label, nbytes / min(ts) / 1e9, nbytes >> 20, reps, min(ts) * 1e3), flush=True)
    return r


qb = NQ * QLEN
for fl in (0, 1, 65):
    enc = timed("arith encode flags=0x%02x" % fl, lambda: eng.arith_encode_host(quals, [fl] * NQ), qb)
    print("    ratio %.3f" % (sum(map(len, enc)) / qb))
    out = timed("arith decode flags=0x%02x" % fl, lambda: eng.cram_uncompress_blocks([(6, e, QLEN) for e in enc]), qb)
    assert out[0][0] == quals[0] and (out[1] == 0).all()
for fl in (0x05, 0xC5, 0x0D):
    enc = timed("nx16 encode flags=0x%02x" % fl, lambda: eng.ransnx16_encode_host(quals, [fl] * NQ), qb)
    print("    ratio %.3f" % (sum(map(len, enc)) / qb))
    out = timed("nx16 decode flags=0x%02x" % fl, lambda: eng.cram_uncompress_blocks([(5, e, QLEN) for e in enc]), qb)
    assert out[0][0] == quals[0] and (out[1] == 0).all()
nb = sum(map(len, nameblocks))
for ua in (0, 1):
    enc = timed("tok3 encode use_arith=%d" % ua, lambda: eng.tok3_encode_host(nameblocks, [ua] * NB), nb, reps=2)
    print("    ratio %.3f" % (sum(map(len, enc)) / nb))
    out = timed("tok3 decode use_arith=%d" % ua, lambda: eng.cram_uncompress_blocks([(8, e, len(d)) for e, d in zip(enc, nameblocks)]), nb)
    assert out[0][0] == nameblocks[0] and (out[1] == 0).all()
