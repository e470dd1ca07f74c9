import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from htslib_amd import _native as nat
from tests.test_tok3 import illumina_names
eng = nat.Engine(0)
names = [illumina_names(np.random.default_rng(i), 10_000) for i in range(8)]
blocks = [names[i % 8] for i in range(64)]
enc = eng.tok3_encode_host(blocks, [0] * 64)
for _ in range(3):
    out = eng.cram_uncompress_blocks([(8, e, len(d)) for e, d in zip(enc, blocks)])
assert out[0][0] == blocks[0]
