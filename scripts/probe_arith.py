import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from htslib_amd import _native as nat
from tests.test_rans4x8 import synth_series
eng = nat.Engine(0)
rng = np.random.default_rng(1)
base = [synth_series(rng, "qual41", 300_000) for _ in range(16)]
quals = [base[i % 16] for i in range(1024)]
for fl in (0, 1):
    enc = eng.arith_encode_host(quals, [fl] * 1024)
    for _ in range(3):
        out = eng.cram_uncompress_blocks([(6, e, 300_000) for e in enc])
    assert out[0][0] == quals[0]
