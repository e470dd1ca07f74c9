"""FETCH_SIZE calibration in OUR access patterns (MI355X_MICROARCH.md: only wide coalesced 16 B/lane streams are known
to read exactly 1/2; everything else must be calibrated on a known byte count).  Pattern A = the CRC pass of the inflate
kernel (hg_crc32_dev: every lane walks its own 1 KiB slice of a 64 KiB block with 16-byte loads), exactly 4 GiB read once.
Run under: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv"""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from htslib_amd import _native as nat
eng = nat.Engine(0)
N, L = 65536, 65536                      # 65536 blocks x 64 KiB = 4 GiB
d = torch.randint(0, 255, (N * L,), dtype=torch.uint8, device="cuda")
off = torch.arange(N, dtype=torch.int64, device="cuda") * L
ln = torch.full((N,), L, dtype=torch.int32, device="cuda")
crc = torch.zeros(N, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    eng.crc32_dev(d.data_ptr(), off.data_ptr(), ln.data_ptr(), N, crc.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("bytes per launch", N * L)
