"""Host-side helpers that stage a BGZF stream in HBM and run the gfx950 batch kernels on it.

torch is used for device memory / streams only; every codec operation goes through the C ABI
(include/htsgpu.h).  Mirrors the reader side of bgzf.c: framing scan on the host
(bgzf_mt_read_block, bgzf.c:1485-1539), block decode + CRC on the device (bgzf_decode_func,
bgzf.c:1373-1384).
"""
from __future__ import annotations

import numpy as np

from . import _native as nat


def shard_blocks(desc, world: int):
    """Static split of a block list over `world` GPUs: contiguous ranges of blocks,
    balanced by uncompressed bytes, no exchange.  Returns [(first, last_exclusive)] per rank."""
    n = len(desc)
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)
    csum = np.cumsum(desc["ulen"].astype(np.int64))
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        cuts.append(int(np.searchsorted(csum, target, side="left")) + (0 if target == 0 else 1))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.minimum(cuts, n)).tolist()
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def reduce_timing(elapsed: float, plain_bytes: float, comp_bytes: float, ok: bool, world: int, device=None):
    """What bench.py reports for N ranks: MAX of the per-rank elapsed time, SUM of bytes, AND of
    the verification flags.  Uses torch.distributed when world > 1 (RCCL on GPUs, gloo in tests)."""
    if world <= 1:
        return elapsed, plain_bytes, comp_bytes, ok
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([plain_bytes, comp_bytes, 0.0 if ok else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(t[0]), float(tot[0]), float(tot[1]), int(tot[2]) == 0


class DeviceStream:
    """A BGZF stream resident in HBM together with its block descriptors."""

    def __init__(self, comp: bytes, device: int = 0):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.desc, self.total_u = nat.bgzf_scan(comp)
        self.nblocks = len(self.desc)
        self.comp_len = len(comp)
        pad = (-self.comp_len) % 256 + 256            # dword-padded as the C ABI requires
        self.d_comp = torch.zeros(self.comp_len + pad, dtype=torch.uint8, device=self.dev)
        if self.comp_len:
            self.d_comp[:self.comp_len].copy_(torch.frombuffer(bytearray(comp), dtype=torch.uint8))
        self.d_desc = torch.from_numpy(self.desc.view(np.uint8).reshape(-1).copy()).to(self.dev)
        self.d_out = torch.zeros(self.total_u + 256, dtype=torch.uint8, device=self.dev)
        self.d_status = torch.full((max(self.nblocks, 1),), 77, dtype=torch.int32, device=self.dev)

    def inflate(self, eng: "nat.Engine", stream: int | None = None):
        s = self.torch.cuda.current_stream().cuda_stream if stream is None else stream
        eng.bgzf_inflate_dev(self.d_comp.data_ptr(), self.comp_len, self.d_desc.data_ptr(), self.nblocks,
                             self.d_out.data_ptr(), self.total_u, self.d_status.data_ptr(), s)

    def result(self):
        """(plain bytes, per-block status ndarray) after synchronising."""
        self.torch.cuda.synchronize()
        return (self.d_out[:self.total_u].cpu().numpy().tobytes(),
                self.d_status[:self.nblocks].cpu().numpy())


def crc32_device(eng: "nat.Engine", buffers, device: int = 0):
    """CRC-32 of each bytes object in `buffers`, computed by hg_crc32_dev."""
    import torch
    dev = torch.device("cuda", device)
    lens = np.array([len(b) for b in buffers], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))[:-1]]).astype(np.uint64) if len(lens) else np.zeros(0, np.uint64)
    blob = b"".join(buffers)
    d_data = torch.zeros(len(blob) + 256, dtype=torch.uint8, device=dev)
    if blob:
        d_data[:len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    d_off = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.view(np.int32)).to(dev)
    d_crc = torch.zeros(max(len(lens), 1), dtype=torch.int32, device=dev)
    eng.crc32_dev(d_data.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), len(lens), d_crc.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_crc[:len(lens)].cpu().numpy().view(np.uint32)
