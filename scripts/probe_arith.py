"""GPU box: per-symbol time of the range coder's chains (host API, streams all alike so the launch lasts one chain).  python scripts/probe_arith.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from htslib_amd import _native as nat
from tests.test_rans4x8 import synth_series
eng = nat.Engine(0)
rng = np.random.default_rng(1)
cases = [("random bytes, 40 KB", lambda: rng.integers(0, 256, 40_000, dtype=np.uint8).tobytes(), 40_000),
         ("qual41, 300 KB", lambda: synth_series(rng, "qual41", 300_000), 300_000),
         ("qual4, 300 KB", lambda: synth_series(rng, "qual4", 300_000), 300_000)]
for name, gen, n in cases:
    base = [gen() for _ in range(8)]
    data = [base[i % 8] for i in range(512)]
    for fl in (0, 1, 64, 65):
        t = time.perf_counter(); enc = eng.arith_encode_host(data, [fl] * 512); te = time.perf_counter() - t
        t = time.perf_counter(); enc = eng.arith_encode_host(data, [fl] * 512); te = min(te, time.perf_counter() - t)
        blocks = [(6, e, n) for e in enc]
        out, st = eng.cram_uncompress_blocks(blocks)
        t = time.perf_counter(); out, st = eng.cram_uncompress_blocks(blocks); td = time.perf_counter() - t
        assert (st == 0).all() and out[0] == data[0]
        print("%-22s flags %2d: encode %6.1f ms (%5.0f ns/symbol)  decode %6.1f ms (%5.0f ns/symbol)  ratio %.3f" % (name, fl, te * 1e3, te / n * 1e9, td * 1e3, td / n * 1e9, len(enc[0]) / n), flush=True)
