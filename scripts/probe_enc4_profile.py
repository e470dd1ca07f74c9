"""Phase times of ransnx16_encode_kernel<4> on the small 4-way streams of a slice batch (a -DHG_ENC_PROFILE build of ransnx16_enc.hip: build/prof/libhtsgpu_prof.so;
ticks of 10 ns).  GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ["HTSGPU_LIB"] = os.path.abspath(os.path.join("build", "prof", "libhtsgpu_prof.so"))
from htslib_amd import _native as nat
eng = nat.Engine(0)
rng = np.random.default_rng(1)
for name, d in (("bytes 40k flat", rng.integers(0, 256, 40_000).astype(np.uint8)), ("40 levels 40k", rng.integers(0, 40, 40_000).astype(np.uint8)), ("4 levels 20k", rng.integers(0, 4, 20_000).astype(np.uint8))):
    for fl in (0, 1):
        print("==", name, "flags", fl, flush=True)
        eng.ransnx16_encode_host([bytes(np.roll(d, i)) for i in range(32)], [fl] * 32)
