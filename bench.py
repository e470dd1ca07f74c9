#!/usr/bin/env python3
"""bench.py -- BGZF inflate of a synthetic BAM on MI355X (BASELINE.json configs[1]).

A "step" is ONE pass of the hot path over the whole workload: every BGZF block
of the (per-GPU) synthetic BAM is inflated + CRC-checked by one launch of the
gfx950 kernel, inputs and outputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gib G] [--op all|inflate|deflate|rans|cram|bam|e2e]

The default (--op all) prints ONE JSON line: the headline is BGZF inflate (BASELINE configs[1]) and `extra` carries
the other quantities of BASELINE.json's metric -- BGZF deflate (configs[2]), CRAM rANS Nx16 decode (configs[3]),
whole CRAM 3.1 slice encode/decode (configs[4] shape) -- each with its own `roofline` and `cpu_baseline`, plus
`end_to_end`: what a libhts caller sees through bgzf_read / bgzf_write (file I/O + PCIe + kernels, overlapped) and
the latency of random bgzf_seek + read, next to the reference library doing the same on the host cores.

N>1 is launched by the driver through torch.distributed.run (one rank per GPU,
blocks are independent so there is NO data-path collective; the only collective
is the barrier / max-reduce that brackets the timed region).  Weak scaling:
every rank inflates its own G GiB (different seeds).

The JSON line carries `roofline` (HBM bound, algorithmic bytes C+U per block,
SURVEY.md 8d) and, at N=1, `cpu_baseline` = the REAL reference (oracle/_ref,
htslib bgzf.c + libdeflate, its own thread pool) timed on this host's cores.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import multiprocessing as mp
import os
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the library asks for 16 hardware queues when it is loaded (htsgpu_api.hip); torch initialises HIP first in this process, so the request is made here as well
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

CHUNK = 32 << 20          # plain bytes generated + deflated by one worker task
GEN_VERSION = "v1"
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def _prep_chunk(args):
    """Worker: synthesise chunk `idx` and deflate it the way stock htslib (zlib, level 6) does."""
    seed, idx, nbytes, level, cache_dir = args
    from htslib_amd import synth
    key = f"{GEN_VERSION}{'' if synth.QUAL_VARIANT == 'A' else synth.QUAL_VARIANT}_{seed:x}_{idx}_{nbytes}_{level}"
    path = os.path.join(cache_dir, key + ".bgzf") if cache_dir else None
    if path and os.path.exists(path):
        with open(path, "rb") as f:
            bg = f.read()
        return idx, bg, None
    data, bg = synth.bam_bgzf(nbytes, seed=seed, chunk=idx, level=level, threads=1,
                              with_header=(idx == 0), eof=False)
    if path:
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(bg)
        os.replace(tmp, path)
    return idx, bg, len(data)


def prepare(seed: int, total_bytes: int, level: int, workers: int, cache_dir: str | None):
    """Returns the BGZF stream (bytes) of >= total_bytes of synthetic BAM."""
    nchunks = max(1, (total_bytes + CHUNK - 1) // CHUNK)
    per = min(CHUNK, total_bytes) if nchunks == 1 else CHUNK
    tasks = [(seed, i, per, level, cache_dir) for i in range(nchunks)]
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
    parts = [None] * nchunks
    if workers > 1 and nchunks > 1:
        ctx = mp.get_context("fork")
        with ctx.Pool(min(workers, nchunks)) as pool:
            for idx, bg, _ in pool.imap_unordered(_prep_chunk, tasks):
                parts[idx] = bg
    else:
        for t in tasks:
            idx, bg, _ = _prep_chunk(t)
            parts[idx] = bg
    return b"".join(parts)


def prepare_shared(rank: int, world: int, seed: int, total_bytes: int, level: int, workers: int, cache_dir: str, rotate: bool = True,
                   timeout_s: float = 3600.0):
    """Multi-GPU runs: ONE synthetic data set is generated cooperatively (rank r builds chunks r, r+world, ... into the
    shared cache, every rank uses all of its worker processes) and each rank takes the whole set starting at its own
    chunk -- its own shard order of independent BGZF blocks -- instead of every rank synthesising and deflating a private
    10 GiB.  Returns (stream, index of the first chunk)."""
    nchunks = max(1, (total_bytes + CHUNK - 1) // CHUNK)
    per = min(CHUNK, total_bytes) if nchunks == 1 else CHUNK
    os.makedirs(cache_dir, exist_ok=True)
    mine = [(seed, i, per, level, cache_dir) for i in range(nchunks) if i % world == rank]
    if mine:
        if workers > 1 and len(mine) > 1:
            ctx = mp.get_context("fork")
            with ctx.Pool(min(workers, len(mine))) as pool:
                for _ in pool.imap_unordered(_prep_chunk, mine):
                    pass
        else:
            for t in mine:
                _prep_chunk(t)
    paths = [os.path.join(cache_dir, f"{GEN_VERSION}_{seed:x}_{i}_{per}_{level}.bgzf") for i in range(nchunks)]
    t0 = time.time()
    for pth in paths:                                       # the other ranks' chunks appear atomically (tmp + rename)
        while not os.path.exists(pth):
            if time.time() - t0 > timeout_s:
                raise RuntimeError("timed out waiting for " + pth)
            time.sleep(0.05)
    start = (rank * nchunks) // world if rotate else 0
    parts = []
    for k in range(nchunks):
        with open(paths[(start + k) % nchunks], "rb") as f:
            parts.append(f.read())
    return b"".join(parts), start


def cpu_baseline(bgzf_sample: bytes, plain_len: int, threads: int):
    """Reference htslib (bgzf.c + libdeflate 1.8, hts_tpool) decoding the sample: ref_bgzip -d -@T."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bgzip_ld")
    if not os.path.exists(exe):
        return None
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(shm, f"htsgpu_cpu_baseline_{os.getpid()}.bam.gz")
    with open(path, "wb") as f:
        f.write(bgzf_sample)
        from htslib_amd import synth
        f.write(synth.BGZF_EOF)
    times = []
    try:
        for _ in range(3):
            t = time.perf_counter()
            with open(os.devnull, "wb") as dn:
                r = subprocess.run([exe, "-d", "-c", "-@", str(threads), path], stdout=dn, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t
            if r.returncode != 0:
                return None
            times.append(dt)
    finally:
        os.unlink(path)
    best = sorted(times)[1]                                   # median of 3
    return {"value": round(plain_len / best / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "reference",
            "sample": f"oracle/_ref/ref_bgzip_ld -d -@{threads} (htslib bgzf.c + libdeflate 1.8, hts_tpool) on "
                      f"{plain_len / 2**30:.2f} GiB of the same BAM from /dev/shm, median of 3"}



def cpu_baseline_deflate(sample: bytes, level: int, threads: int):
    """Reference htslib (bgzf.c + libdeflate 1.8, hts_tpool) compressing the sample: ref_bgzip -l L -@T; median of 3."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bgzip_ld")
    if not os.path.exists(exe):
        return None
    p = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"htsgpu_defl_in_{os.getpid()}")
    open(p, "wb").write(sample)
    times = []
    try:
        for _ in range(3):
            t = time.perf_counter()
            with open(os.devnull, "wb") as dn:
                r = subprocess.run([exe, "-c", "-l", str(level), "-@", str(threads), p], stdout=dn)
            if r.returncode != 0:
                return None
            times.append(time.perf_counter() - t)
    finally:
        os.unlink(p)
    return {"value": round(len(sample) / sorted(times)[1] / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "reference",
            "sample": f"oracle/_ref/ref_bgzip_ld -l{level} -@{threads} (htslib bgzf.c + libdeflate 1.8) on "
                      f"{len(sample) / 2**30:.2f} GiB of the same BAM from /dev/shm, median of 3"}


def hbm_traffic_file(name, key_sum, plain_bytes):
    """HBM bytes per launch from a committed PMC summary under profiles/ (per plain byte, scaled to this workload)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", name)))
        return int(sum(t[k] for k in key_sum) * plain_bytes)
    except Exception:
        return None


def symbols_per_byte(comp: bytes, desc, nblocks: int = 64):
    """Huffman symbols (literals + matches) per plain byte, counted by the oracle's inflate on a sample of blocks."""
    import ctypes as C
    try:
        orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        orc.orc_bgzf_decompress_stream.restype = C.c_long
        orc.orc_bgzf_decompress_stream.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        orc.orc_symbol_counts.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]
        step = max(1, len(desc) // nblocks)
        lit, mat = C.c_ulonglong(), C.c_ulonglong()
        orc.orc_symbol_counts(None, None, 1)
        plain = 0
        buf = C.create_string_buffer(65536 + 64)
        for i in range(0, len(desc), step):
            d = desc[i]
            blk = comp[int(d["coff"]):int(d["coff"]) + int(d["clen"])]
            n = orc.orc_bgzf_decompress_stream(blk, len(blk), buf, 65536 + 64)
            if n > 0:
                plain += n
        orc.orc_symbol_counts(C.byref(lit), C.byref(mat), 1)
        return (lit.value + mat.value) / max(plain, 1), lit.value / max(lit.value + mat.value, 1)
    except Exception:
        return None, None


class Run:
    """Process-wide bench context: rank / world / device, torch.distributed when world > 1."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        # Rehearsal of the multi-rank path on a box with ONE GPU (tests/test_multigpu_gpu.py): every rank takes device 0 and the ranks meet over gloo --
        # RCCL refuses two ranks on one device.  Everything else (rank environment, sharding, per-rank verification, MAX-over-ranks timing) is the real path.
        self.share_device = os.environ.get("HTS_BENCH_SHARE_DEVICE") == "1"
        if self.share_device:
            self.local = 0
        if self.world > 1:
            args.gpus = self.world
        self.ncores = os.cpu_count() or 1
        self.workers = args.workers or max(1, (self.ncores - 4) // max(1, self.world))
        self.dist = None
        self.dev = None

    def init_device(self):
        import torch
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        if self.world > 1 and self.dist is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.share_device: dist.init_process_group("gloo")
            else: dist.init_process_group("nccl", device_id=self.dev)
            self.dist = dist
        return torch

    @property
    def reduce_dev(self):
        """where the tensors of a cross-rank reduction live: the rank's GPU (RCCL), the host in the one-device rehearsal (gloo)"""
        return None if self.share_device else self.dev

    def barrier(self):
        import torch
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def finish(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None

    def timed(self, step, steps, warmup):
        """W untimed warm-ups, then EXACTLY `steps` steps bracketed by barrier + synchronize; per-step HIP events."""
        import torch
        for _ in range(warmup):
            step()
        self.barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in evs:
            a.record(); step(); b.record()
        self.barrier()
        elapsed = time.perf_counter() - t0
        return elapsed, float(np.mean([a.elapsed_time(b) for a, b in evs]))


class Staged:
    """The synthetic BAM of this rank, compressed (host bytes + HBM) and the buffers of the inflate launch."""


def stage(run: Run) -> Staged:
    args = run.args
    S = Staged()
    total_bytes = int(args.gib * (1 << 30))
    cache = None if args.no_cache else os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "htsgpu_bench_cache")
    t0 = time.perf_counter()
    S.first_chunk = 0
    strong = args.scaling == "strong" and run.world > 1
    if run.world > 1 and cache:
        S.seed = 0x5EED0001
        comp, S.first_chunk = prepare_shared(run.rank, run.world, S.seed, total_bytes, args.level, run.workers, cache,
                                             rotate=(args.op != "bam" and not strong))
    else:
        S.seed = 0x5EED0001 + (0 if strong else 1000003 * run.rank)
        comp = prepare(S.seed, total_bytes, args.level, run.workers, cache)
    S.t_prep = time.perf_counter() - t0
    torch = run.init_device()
    from htslib_amd import _native as nat
    from htslib_amd.bgzf import shard_blocks
    S.eng = nat.Engine(run.local)
    desc, total_u = nat.bgzf_scan(comp)
    S.whole_blocks, S.whole_plain = len(desc), int(total_u)
    S.shard = None
    if strong:
        # ONE data set, split by block ranges balanced on plain bytes; rank r decodes only its range
        lo, hi = shard_blocks(desc, run.world)[run.rank]
        S.shard = (lo, hi)
        c0 = int(desc["coff"][lo]) if lo < len(desc) else len(comp)
        c1 = int(desc["coff"][hi - 1] + desc["clen"][hi - 1]) if hi > lo else c0
        comp = comp[c0:c1]
        desc, total_u = nat.bgzf_scan(comp)
    S.comp, S.desc, S.total_u = comp, desc, int(total_u)
    S.nblocks, S.comp_len = len(desc), len(comp)
    pad = (-S.comp_len) % 256 + 256
    S.d_comp = torch.zeros(S.comp_len + pad, dtype=torch.uint8, device=run.dev)
    S.d_comp[:S.comp_len].copy_(torch.frombuffer(bytearray(comp), dtype=torch.uint8))
    S.d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(run.dev)
    S.d_out = torch.empty(S.total_u + 256, dtype=torch.uint8, device=run.dev)
    S.d_status = torch.full((max(S.nblocks, 1),), 77, dtype=torch.int32, device=run.dev)
    torch.cuda.synchronize()
    S.stream = torch.cuda.current_stream().cuda_stream
    return S


def inflate_variant_b(run: Run, steps: int, gib: float = 2.0):
    """SURVEY 8d variant B beside the headline: the same BAM generator with HiSeq-like 41-level qualities (ratio ~3.3 instead of ~5.5; many more
    literals per byte).  GB/s flatters easy data, symbols/s is the unit to compare."""
    import torch
    from htslib_amd import synth, _native as nat
    synth.QUAL_VARIANT = "B"
    try:
        cache = None if run.args.no_cache else os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "htsgpu_bench_cache")
        comp = prepare(0x5EED0B01, int(gib * (1 << 30)), run.args.level, run.workers, cache)
        plain0, _, _ = synth.bam_stream(min(CHUNK, int(gib * (1 << 30))), 0x5EED0B01, 0, True)
    finally:
        synth.QUAL_VARIANT = "A"
    eng = nat.Engine(run.local)
    desc, total_u = nat.bgzf_scan(comp)
    d_comp = torch.zeros(len(comp) + 512, dtype=torch.uint8, device=run.dev)
    d_comp[:len(comp)].copy_(torch.frombuffer(bytearray(comp), dtype=torch.uint8))
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(run.dev)
    d_out = torch.empty(int(total_u) + 256, dtype=torch.uint8, device=run.dev)
    d_status = torch.full((len(desc),), 77, dtype=torch.int32, device=run.dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.bgzf_inflate_dev(d_comp.data_ptr(), len(comp), d_desc.data_ptr(), len(desc), d_out.data_ptr(), int(total_u), d_status.data_ptr(), stream)

    elapsed, k_ms = run.timed(step, steps, 1)
    ok = int((d_status != 0).sum()) == 0 and d_out[:len(plain0)].cpu().numpy().tobytes() == plain0
    spb, litfrac = symbols_per_byte(comp, desc)
    gbs = total_u * steps / elapsed / 1e9
    return {"metric": "BGZF inflate, variant B (41-level qualities), uncompressed GB/s", "value": round(gbs, 3), "unit": "GB/s", "steps": steps, "ms_per_step": round(elapsed * 1e3 / steps, 3),
            "ratio": round(total_u / len(comp), 3), "plain_bytes": int(total_u), "symbols_per_plain_byte": None if spb is None else round(spb, 4),
            "literal_fraction_of_symbols": None if litfrac is None else round(litfrac, 3), "Gsymbols_per_s": None if spb is None else round(gbs * spb, 2),
            "roofline_frac": round((total_u + len(comp)) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "kernel_ms": round(k_ms, 3), "verified": bool(ok)}


# ---- the headline on PRODUCTION-flavour input: blocks as an htslib built with libdeflate writes them (INSTALL:41, bgzf.c:561-616) ----------------------
_LD = {"comp": None, "desc": None}
LIBDEFLATE_CANDIDATES = ("/opt/conda/lib/libdeflate.so.0", os.path.join(ROOT, "oracle", "_ref", "ld", "libdeflate.so.0"), "libdeflate.so.0")


def _libdeflate():
    import ctypes as C
    for n in LIBDEFLATE_CANDIDATES:
        try:
            L = C.CDLL(n)
        except OSError:
            continue
        L.libdeflate_alloc_compressor.restype = C.c_void_p; L.libdeflate_alloc_compressor.argtypes = [C.c_int]
        L.libdeflate_deflate_compress.restype = C.c_size_t
        L.libdeflate_deflate_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.libdeflate_free_compressor.argtypes = [C.c_void_p]
        return L
    return None


def _ld_worker(rng):
    """blocks [a, b) of the zlib-flavour stream -> the same plain blocks as bgzf_compress's libdeflate branch frames them (level 6 -> libdeflate 7, bgzf.c:583-612)"""
    import ctypes as C, zlib
    a, b, level = rng
    L = _libdeflate()
    comp, desc = _LD["comp"], _LD["desc"]
    z = L.libdeflate_alloc_compressor([0, 1, 2, 3, 5, 6, 7, 8, 10, 12][level])
    buf = C.create_string_buffer(65536)
    out = []
    hdr = bytes.fromhex("1f8b08040000000000ff0600424302 00".replace(" ", ""))
    for i in range(a, b):
        co, cl = int(desc["coff"][i]), int(desc["clen"][i])
        plain = zlib.decompress(comp[co + 18:co + cl - 8], -15)
        n = L.libdeflate_deflate_compress(z, plain, len(plain), buf, 65536 - 26)
        assert n > 0
        out.append(hdr + struct.pack("<H", n + 25) + buf.raw[:n] + struct.pack("<II", zlib.crc32(plain), len(plain)))
    L.libdeflate_free_compressor(z)
    return a, b"".join(out)


def inflate_libdeflate(run: Run, comp: bytes, desc, steps: int, level: int):
    """BASELINE configs[1] on blocks as PRODUCTION htslib writes them: the headline's plain blocks re-compressed by libdeflate exactly as bgzf_compress's
    HAVE_LIBDEFLATE branch does (several deflate blocks per BGZF block, denser streams than zlib's).  Same kernel, same launch, same verification."""
    import torch, zlib
    from htslib_amd import _native as nat
    if _libdeflate() is None:
        return {"error": "no libdeflate on this host"}
    t0 = time.time()
    _LD["comp"], _LD["desc"] = comp, desc
    nb = len(desc)
    per = max(64, (nb + run.workers * 8 - 1) // max(1, run.workers * 8))
    tasks = [(a, min(nb, a + per), level) for a in range(0, nb, per)]
    parts = {}
    if run.workers > 1:
        with mp.get_context("fork").Pool(run.workers) as pool:
            for a, bg in pool.imap_unordered(_ld_worker, tasks):
                parts[a] = bg
    else:
        for t in tasks:
            a, bg = _ld_worker(t); parts[a] = bg
    _LD["comp"] = _LD["desc"] = None
    ld = b"".join(parts[a] for a in sorted(parts))
    del parts
    t_prep = time.time() - t0
    eng = nat.Engine(run.local)
    d2, total_u = nat.bgzf_scan(ld)
    assert len(d2) == nb
    d_comp = torch.zeros(len(ld) + 512, dtype=torch.uint8, device=run.dev)
    d_comp[:len(ld)].copy_(torch.frombuffer(bytearray(ld), dtype=torch.uint8))
    d_desc = torch.from_numpy(d2.view(np.uint8).reshape(-1).copy()).to(run.dev)
    d_out = torch.empty(int(total_u) + 256, dtype=torch.uint8, device=run.dev)
    d_status = torch.full((nb,), 77, dtype=torch.int32, device=run.dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.bgzf_inflate_dev(d_comp.data_ptr(), len(ld), d_desc.data_ptr(), nb, d_out.data_ptr(), int(total_u), d_status.data_ptr(), stream)

    elapsed, k_ms = run.timed(step, steps, 1)
    # every block's CRC-32 was checked in-kernel; the first 32 MiB are also compared with an independent inflate (Python zlib) of the same blocks
    ok = int((d_status != 0).sum()) == 0
    ncheck = int(np.searchsorted(d2["uoff"], 32 << 20))
    ref = b"".join(zlib.decompress(ld[int(d2["coff"][i]) + 18:int(d2["coff"][i]) + int(d2["clen"][i]) - 8], -15) for i in range(ncheck))
    ok = ok and d_out[:len(ref)].cpu().numpy().tobytes() == ref
    spb, litfrac = symbols_per_byte(ld, d2)
    gbs = total_u * steps / elapsed / 1e9
    # deflate blocks per BGZF block (libdeflate splits; zlib-6 writes one): sampled by walking the headers is the oracle's job (symbols_per_byte); here: ratio only
    return {"metric": "BGZF inflate, blocks written by libdeflate (production htslib flavour, level %d), uncompressed GB/s" % level, "value": round(gbs, 3), "unit": "GB/s", "steps": steps,
            "ms_per_step": round(elapsed * 1e3 / steps, 3), "ratio": round(total_u / len(ld), 3), "plain_bytes": int(total_u), "compressed_bytes": len(ld), "blocks": nb,
            "symbols_per_plain_byte": None if spb is None else round(spb, 4), "literal_fraction_of_symbols": None if litfrac is None else round(litfrac, 3),
            "Gsymbols_per_s": None if spb is None else round(gbs * spb, 2),
            "roofline": {"bound": "hbm", "achieved": round((total_u + len(ld)) / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round((total_u + len(ld)) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "kernel": "hg::bgzf_inflate_kernel", "kernel_ms": round(k_ms, 3),
                         "algorithmic_bytes_per_launch": int(total_u + len(ld))},
            "prep_seconds": round(t_prep, 1), "verified": bool(ok),
            "note": "same plain blocks as the headline, re-compressed with libdeflate level map 6 -> 7 exactly as bgzf.c:583-612 frames them"}


def host_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown.  The round-6 GPU box shows 256 host threads and
    grants 16: every cpu_baseline of this file -- the reference's thread pools as much as the 254-process codec ports, whose workers start seconds apart (each is handed a
    12-14 MB pickled sample) and therefore mostly measure an unshared core each -- has to be read with that in mind (profiles/r06_host_scaling_probe.txt)."""
    try:
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            return None if q == "max" else round(int(q) / int(p_), 2)
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p_, 2)
    except Exception:
        return None


def op_inflate(run: Run, S: Staged, steps: int, warmup: int):
    import torch
    from htslib_amd import synth
    from htslib_amd.bgzf import reduce_timing
    args = run.args

    def step():
        S.eng.bgzf_inflate_dev(S.d_comp.data_ptr(), S.comp_len, S.d_desc.data_ptr(), S.nblocks, S.d_out.data_ptr(), S.total_u,
                               S.d_status.data_ptr(), S.stream)

    elapsed, k_ms = run.timed(step, steps, warmup)
    # ---------------- verification (outside the timed region) ------------------------------
    nbad = int((S.d_status[:S.nblocks] != 0).sum()) if S.nblocks else 0
    ok = nbad == 0
    strong = S.shard is not None
    # every block's CRC-32 (written by the host deflater over the ORIGINAL bytes) was re-checked in-kernel on every rank;
    # additionally the first chunk of the stream is compared byte for byte with a regenerated plain image
    if not strong or run.rank == 0:
        chk = min(CHUNK, int(args.gib * (1 << 30)), S.total_u)
        plain0, _, _ = synth.bam_stream(min(CHUNK, int(args.gib * (1 << 30))), S.seed, S.first_chunk, S.first_chunk == 0)
        chk = min(chk, len(plain0))
        ok = ok and S.d_out[:chk].cpu().numpy().tobytes() == plain0[:chk]
    elapsed, sum_u, sum_c, ok = reduce_timing(elapsed, float(S.total_u), float(S.comp_len), ok, run.world, run.reduce_dev)
    if strong:
        ok = ok and int(sum_u) == S.whole_plain                        # the shards cover the file exactly once
    if run.rank != 0:
        return None, ok
    value = sum_u * steps / elapsed / 1e9
    alg_bytes = float(S.total_u + S.comp_len)                       # per launch on this rank: C + U
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    spb, litfrac = symbols_per_byte(S.comp, S.desc)
    out = {
        "metric": "BGZF inflate throughput, uncompressed GB/s (decode, CRC-checked, HBM-resident)",
        "value": round(value, 3), "unit": "GB/s", "n_gpus": run.world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed * 1e3 / steps, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": (f"BGZF inflate of ONE {args.gib:g} GiB synthetic coordinate-sorted 150bp BAM split over the GPUs by block ranges "
                                if strong else f"BGZF inflate of a {args.gib:g} GiB synthetic coordinate-sorted 150bp BAM per GPU ") +
                               f"(zlib level {args.level} blocks as written by stock htslib, <=65280 B each)",
                   "blocks_per_gpu": S.nblocks, "plain_bytes_per_gpu": int(S.total_u),
                   "compressed_bytes_per_gpu": int(S.comp_len), "ratio": round(S.total_u / max(S.comp_len, 1), 3),
                   "symbols_per_plain_byte": None if spb is None else round(spb, 4),
                   "literal_fraction_of_symbols": None if litfrac is None else round(litfrac, 3),
                   "Gsymbols_per_s": None if spb is None else round(value * spb, 2),
                   "sharding": "independent blocks, static split, no collective", "verified": bool(ok),
                   "prep_seconds": round(S.t_prep, 1), "host_cpus": os.cpu_count(), "host_cpu_quota": host_cpu_quota()},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": hbm_traffic_file("hbm_traffic_inflate.json", ["fetch_bytes_per_plain_byte", "write_bytes_per_plain_byte"], S.total_u),
                     "kernel": "hg::bgzf_inflate_kernel", "kernel_ms": round(k_ms, 3),
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "traffic_note": "FETCH_SIZE + WRITE_SIZE PMC passes (profiles/hbm_traffic_inflate.json, with its calibration note)"},
    }
    if run.world == 1 and not args.no_cpu_baseline:
        lim = 4 << 30                                                   # bounded sample: at most 4 GiB plain of the same stream
        if S.total_u > lim:
            cut = int(np.searchsorted(S.desc["uoff"], lim))
            sample, plen = S.comp[:int(S.desc["coff"][cut])], int(S.desc["uoff"][cut])
        else:
            sample, plen = S.comp, int(S.total_u)
        cb = cpu_baseline(sample, plen, run.ncores)
        if cb:
            out["cpu_baseline"] = cb
    return out, ok


def _rans_series(seed, nslices):
    """QS (4-bin Markov qualities) and BA (bases) data series of `nslices` CRAM slices, 10 000 x 150 bp each."""
    out = []
    for sl in range(nslices):
        rng = np.random.Generator(np.random.PCG64(seed + sl))
        n = 10_000 * 150
        change = rng.random(n) < 0.1
        idx = np.maximum.accumulate(np.where(change, np.arange(n), 0))
        qs = np.array([2, 12, 23, 37], dtype=np.uint8)[rng.integers(0, 4, n)][idx]
        ba = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.choice(5, n, p=[.2495, .2495, .2495, .2495, .002])]
        out.append((qs.tobytes(), ba.tobytes()))
    return out



def _rans_cpu_worker(task):
    """cpu_baseline worker: the oracle's scalar Nx16 decoder on a share of the streams (one process per core)."""
    import ctypes as C
    streams, reps = task
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    orc.orc_ransnx16_uncompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    buf = C.create_string_buffer(1_500_000 + 64); got = C.c_size_t(0)
    done = 0
    t = time.perf_counter()
    for _ in range(reps):
        for s in streams:
            orc.orc_ransnx16_uncompress(s, len(s), buf, 1_500_064, C.byref(got)); done += got.value
    return done, time.perf_counter() - t                       # the worker times its own loop (pool start-up excluded)


def _cram_cpu_worker(task):
    """cpu_baseline worker for the CRAM block codecs: the oracle's scalar C restatements (NOT reference code) on a share of blocks.  mode "dec": each
    (method, stream, plaintext) is decoded; "enc": each plaintext is encoded with the codec family that won on the GPU (one call = the tuner's steady state)."""
    import ctypes as C, zlib
    blocks, mode, seconds = task
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    sz = C.c_size_t
    for f in ("orc_ransnx16_uncompress", "orc_rans4x8_uncompress", "orc_tok3_decode"):
        getattr(orc, f).argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(sz)]
    orc.orc_arith_uncompress.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(sz), C.c_long]
    for f in ("orc_ransnx16_compress", "orc_arith_compress", "orc_tok3_encode"):
        getattr(orc, f).restype = sz; getattr(orc, f).argtypes = [C.c_char_p, sz, C.c_char_p, C.c_int]
    orc.orc_rans4x8_compress.restype = sz; orc.orc_rans4x8_compress.argtypes = [C.c_char_p, sz, C.c_char_p, C.c_int]
    cap = max(len(b[2]) for b in blocks) * 2 + (1 << 20)
    buf = C.create_string_buffer(cap); got = sz(0)
    done = 0
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        for meth, comp, plain in blocks:
            if mode == "dec":
                if meth == 0: pass
                elif meth == 1: zlib.decompress(comp, 31)
                elif meth == 4: orc.orc_rans4x8_uncompress(comp, len(comp), buf, cap, C.byref(got))
                elif meth == 5: orc.orc_ransnx16_uncompress(comp, len(comp), buf, cap, C.byref(got))
                elif meth == 6: orc.orc_arith_uncompress(comp, len(comp), buf, cap, C.byref(got), -1)
                elif meth == 8: orc.orc_tok3_decode(comp, len(comp), buf, cap, C.byref(got))
            else:
                if meth == 1: zlib.compress(plain, 5)
                elif meth == 4: orc.orc_rans4x8_compress(plain, len(plain), buf, comp[0] & 1)
                elif meth == 5: orc.orc_ransnx16_compress(plain, len(plain), buf, comp[0])
                elif meth == 6: orc.orc_arith_compress(plain, len(plain), buf, comp[0])
                elif meth == 8: orc.orc_tok3_encode(plain, len(plain), buf, comp[8] if len(comp) > 8 else 0)
            done += len(plain)
    return done, time.perf_counter() - t


def cpu_baseline_cram(sample, nproc, mode, seconds=8.0):
    with mp.get_context("fork").Pool(nproc) as pool:
        parts = pool.map(_cram_cpu_worker, [(sample, mode, seconds)] * nproc)
    return {"value": round(sum(d / t for d, t in parts) / 1e9, 3), "unit": "GB/s", "cores": nproc, "kind": "port",
            "sample": "NOT reference code (htscodecs is absent): the oracle's scalar C codecs (oracle/*_oracle.c; zlib for method 1) %s the blocks of four of the same slices "
                      "in a loop for %.0f s on %d processes" % ("decoding" if mode == "dec" else "encoding (each block with the codec and flags that won on the GPU)", seconds, nproc)}


def _rans_variants(eng, qs_list, steps):
    """SURVEY 8d's matrix for the Nx16 codec: 4-way and 32-way x order 0 / 1 x +-PACK x +-RLE, each on the same quality series (4 symbols, runs: both
    transforms apply), encoded by the gfx950 encoder and decoded through hg_ransnx16_decode_host -- the HOST entry point, so these figures
    include the PCIe transfers and the header planning on the host (the headline above is device-resident).  Every output is compared."""
    import ctypes as C
    from htslib_amd import _native as nat
    out = {}
    n = len(qs_list)
    total = sum(len(q) for q in qs_list)
    for x32 in (0, 4):
        for order in (0, 1):
            for xf, tag in ((0, ""), (0x80, "+PACK"), (0x40, "+RLE"), (0xC0, "+PACK+RLE")):
                fl = x32 | order | xf
                streams = eng.ransnx16_encode_host(qs_list, [fl] * n)
                ins = [(C.c_char * len(s_)).from_buffer_copy(s_) for s_ in streams]
                outs = [C.create_string_buffer(len(q)) for q in qs_list]
                ip = (C.c_void_p * n)(*[C.addressof(x) for x in ins]); opp = (C.c_void_p * n)(*[C.addressof(x) for x in outs])
                il = np.array([len(s_) for s_ in streams], np.uint32); ol = np.array([len(q) for q in qs_list], np.uint32); st = np.zeros(n, np.int32)
                ts = []
                for _ in range(steps + 1):
                    t = time.perf_counter()
                    rc = nat.lib.hg_ransnx16_decode_host(eng._h, ip, il.ctypes.data, n, opp, ol.ctypes.data, st.ctypes.data)
                    ts.append(time.perf_counter() - t)
                good = rc == 0 and not st.any() and all(outs[i].raw == qs_list[i] for i in range(n))
                out["%s o%d%s" % ("32-way" if x32 else "4-way", order, tag)] = {
                    "GBps_host_api": round(total / sorted(ts[1:])[len(ts[1:]) // 2] / 1e9, 3), "ratio": round(float(il.sum()) / total, 4), "flags_written": [hex(s_[0]) for s_ in streams[:1]][0],
                    "verified": bool(good)}
    return {"note": "host entry point incl. PCIe + host planning, %d quality series of 1.5 MB each, median of %d; decode of streams written by the gfx950 encoder (format parity with htscodecs UNPINNED)" % (n, steps),
            "results": out}


def op_rans(run: Run, steps: int, warmup: int, slices: int, nway: int = 32):
    """BASELINE configs[3]: CRAM 3.1 rANS Nx16 decode (order-1 QS + order-0 BA, 32-way -- or 4-way, north_star's "rANS 4x16", with nway = 4) of
    `slices` x 10 000 reads per GPU.  The streams are produced by the gfx950 ENCODER (htscodecs is absent: format parity UNPINNED); the timed
    region is the decode launch, device resident."""
    torch = run.init_device()
    from htslib_amd import _native as nat
    from htslib_amd.bgzf import reduce_timing
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(min(64, run.workers)) as pool:
        chunks = pool.starmap(_rans_series, [(0x5EED0001 + 7_000_003 * run.rank + 1000 * i, 1) for i in range(slices)])
    series = [c[0] for c in chunks]
    t_prep = time.perf_counter() - t0
    eng = nat.Engine(run.local)
    plains, flags = [], []
    for qs, ba in series:
        plains += [qs, ba]; flags += [5, 4] if nway == 32 else [1, 0]   # QS: order 1, BA: order 0; X32 or 4-way
    if os.environ.get("HG_BENCH_RANS_FLAGS"):                       # probe: both series with these flags, e.g. "0,0" = everything order 0
        fq, fb = (int(x, 0) for x in os.environ["HG_BENCH_RANS_FLAGS"].split(","))
        flags = [fq, fb] * (len(flags) // 2)
    streams = []
    enc_all = None
    if nway == 4 and run.world == 1:
        # the ENCODE side of the same workload, all streams in ONE call of the host entry point (PCIe both ways included; the buffers are laid out
        # before the clock starts): 4-way streams are one chain of n / 4 steps each, so a launch lasts one chain whatever the number of streams
        import ctypes as C
        ne = len(plains)
        ins = [(C.c_char * len(d)).from_buffer_copy(d) for d in plains]
        bound = nat.lib.hg_ransnx16_compress_bound(max(len(d) for d in plains))
        outs = [C.create_string_buffer(bound) for _ in plains]
        ip = (C.c_void_p * ne)(*[C.addressof(x) for x in ins]); opp = (C.c_void_p * ne)(*[C.addressof(x) for x in outs])
        il = np.array([len(d) for d in plains], np.uint32); fl8 = np.array(flags, np.uint8); ol = np.zeros(ne, np.uint32)
        t_e = time.perf_counter()
        nat.check(nat.lib.hg_ransnx16_encode_host(eng._h, ip, il.ctypes.data, fl8.ctypes.data, ne, opp, ol.ctypes.data), "hg_ransnx16_encode_host")
        enc_all = time.perf_counter() - t_e
        streams = [outs[i].raw[:int(ol[i])] for i in range(ne)]
        del ins, outs
    else:
        for i in range(0, len(plains), 64):                       # encode on the GPU, in batches
            streams += eng.ransnx16_encode_host(plains[i:i + 64], flags[i:i + 64])
    n = len(streams)
    in_len = np.array([len(s) for s in streams], dtype=np.uint32)
    out_len = np.array([len(p) for p in plains], dtype=np.uint32)
    desc = np.zeros(n, dtype=[("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"),
                              ("scratch_off", "<u4"), ("reserved", "<u4")])
    pad = lambda v: (v + 15) & ~np.uint64(15)
    desc["in_len"], desc["out_len"] = in_len, out_len
    desc["in_off"] = np.concatenate([[0], np.cumsum(pad(in_len.astype(np.uint64)))[:-1]])
    desc["out_off"] = np.concatenate([[0], np.cumsum(pad(out_len.astype(np.uint64)))[:-1]])
    words = np.array([512 + min(65792, int(l)) + 272 if f & 1 else 16 for l, f in zip(in_len, flags)], dtype=np.uint64)
    desc["scratch_off"] = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.uint32)
    blob = bytearray(int(desc["in_off"][-1] + pad(np.uint64(in_len[-1]))))
    for d, s_ in zip(desc, streams):
        blob[int(d["in_off"]):int(d["in_off"]) + len(s_)] = s_
    d_in = torch.frombuffer(blob, dtype=torch.uint8).to(run.dev)
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(run.dev)
    total_u = int(out_len.astype(np.uint64).sum()); total_c = int(in_len.astype(np.uint64).sum())
    d_out = torch.zeros(int(desc["out_off"][-1]) + int(out_len[-1]) + 64, dtype=torch.uint8, device=run.dev)
    d_status = torch.full((n,), 77, dtype=torch.int32, device=run.dev)
    d_scratch = torch.zeros(int(words.sum()) + 64, dtype=torch.int32, device=run.dev)
    d_sel = torch.arange(n, dtype=torch.int32, device=run.dev)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()

    def step():
        if nway == 32:
            nat.check(nat.lib.hg_ransnx16_decode_dev(eng._h, d_in.data_ptr(), d_desc.data_ptr(), None, 0, d_sel.data_ptr(), n,
                                                     d_out.data_ptr(), d_status.data_ptr(), d_scratch.data_ptr(), stream), "rans decode")
        else:
            nat.check(nat.lib.hg_ransnx16_decode_dev(eng._h, d_in.data_ptr(), d_desc.data_ptr(), d_sel.data_ptr(), n, None, 0,
                                                     d_out.data_ptr(), d_status.data_ptr(), d_scratch.data_ptr(), stream), "rans decode")

    elapsed, k_ms = run.timed(step, steps, warmup)
    ok = int((d_status != 0).sum()) == 0
    # EVERY stream is compared, on the device (rANS has no checksum: status 0 alone proves little): the plaintexts laid out like the output image
    expect = bytearray(int(d_out.numel()))
    for d, p_ in zip(desc, plains):
        expect[int(d["out_off"]):int(d["out_off"]) + len(p_)] = p_
    d_expect = torch.frombuffer(expect, dtype=torch.uint8).to(run.dev)
    ok = ok and bool(torch.equal(d_out, d_expect))
    del d_expect
    variants = _rans_variants(eng, [qs for qs, _ in series[:24]], max(3, min(steps, 5))) if run.rank == 0 and run.world == 1 and nway == 32 and not getattr(run.args, "no_variants", False) else None
    elapsed, sum_u, sum_c, ok = reduce_timing(elapsed, float(total_u), float(total_c), ok, run.world, run.reduce_dev)
    if run.rank != 0:
        return None, ok
    alg = float(total_u + total_c)
    out = {"metric": "CRAM 3.1 rANS Nx16 decode throughput, uncompressed GB/s (HBM-resident)",
           "value": round(sum_u * steps / elapsed / 1e9, 3), "unit": "GB/s", "n_gpus": run.world, "steps": steps,
           "warmup": warmup, "ms_per_step": round(elapsed * 1e3 / steps, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": f"rANS Nx16 decode of {slices} CRAM slices x 10 000 x 150 bp per GPU: QS (order-1, {nway}-way) "
                                  f"+ BA (order-0, {nway}-way) data series; streams written by the gfx950 encoder; format parity "
                                  "with htscodecs UNPINNED", "streams_per_gpu": n, "plain_bytes_per_gpu": total_u,
                      "compressed_bytes_per_gpu": total_c, "verified": bool(ok), "verified_streams": n, "prep_seconds": round(t_prep, 1),
                      "variants": variants, **({"encode_GBps_host_api_one_call": round(total_u / enc_all / 1e9, 3)} if enc_all else {})},
           "roofline": {"bound": "hbm", "achieved": round(alg / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic": hbm_traffic_file("hbm_traffic_rans.json" if nway == 32 else "hbm_traffic_rans4.json", ["traffic_bytes_per_plain_byte"], total_u),
                        "kernel": "hgn::ransnx16_decode_kernel<32>" if nway == 32 else "hgn::rans4x16_big_decode_kernel", "kernel_ms": round(k_ms, 3),
                        "MBps_per_stream": round(1.5 / (k_ms * 1e-3), 1),
                        "algorithmic_bytes_per_launch": int(alg)}}
    if run.world == 1 and not run.args.no_cpu_baseline:
        # N processes, one share of the streams each ("N threads, one slice each"); ~10-20 s of CPU work
        nproc = max(1, min(run.ncores - 2, n))
        share = [streams[i::nproc][:8] for i in range(nproc)]
        with mp.get_context("fork").Pool(nproc) as pool:
            parts = pool.map(_rans_cpu_worker, [(sh, 2) for sh in share])
        done, dt = sum(p_[0] for p_ in parts), max(p_[1] for p_ in parts)
        out["cpu_baseline"] = {"value": round(done / dt / 1e9, 3), "unit": "GB/s", "cores": nproc, "kind": "port",
                               "sample": f"oracle/ransnx16_oracle.c (scalar C restatement, NOT reference code: htscodecs is absent) decoding "
                                         f"{sum(map(len, share))} of the same streams twice on {nproc} concurrent processes (bytes / slowest worker's loop time)"}
    return out, ok


def op_cram(run: Run, steps: int, slices: int):
    """BASELINE configs[4] shape on the GPUs of one node: every rank encodes (and decodes back) its own `slices` CRAM 3.1
    slices of 10 000 reads -- 8 data series each -- through hg_cram_compress_blocks_metrics_host, i.e. the reference's
    cram_compress_block2 loop with its method auto-tuner, then hg_cram_uncompress_blocks_host.  These are HOST entry
    points (the reference hands the codecs malloc'd blocks), so unlike the other ops the rate includes PCIe both ways."""
    run.init_device()
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_cram_slices
    if run.dist is not None:
        run.dist.barrier()
    r = bench_cram_slices.main(slices, device=run.local, reps=max(5, steps), quiet=True)
    from htslib_amd.bgzf import reduce_timing
    enc_s, sum_u, sum_c, ok = reduce_timing(r["encode_s"], float(r["plain_bytes"]), float(r["comp_bytes"]), True, run.world, run.reduce_dev)
    dec_s, _, _, _ = reduce_timing(r["decode_s"], float(r["plain_bytes"]), float(r["comp_bytes"]), True, run.world, run.reduce_dev)
    if run.rank != 0:
        return None, ok
    # algorithmic bytes of a whole-slice encode through the tuner's steady state: read U, write C (+ one histogram pass: 2U + C)
    alg_enc = 2.0 * r["plain_bytes"] + r["comp_bytes"]
    alg_dec = float(r["plain_bytes"] + r["comp_bytes"])
    return {"metric": "CRAM 3.1 slice encode throughput through the block-method auto-tuner, plain GB/s (host entry points, PCIe included)",
            "value": round(sum_u / enc_s / 1e9, 3), "unit": "GB/s", "n_gpus": run.world, "steps": max(5, steps), "warmup": 1,
            "ms_per_step": round(enc_s * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d slices x 10 000 reads per GPU, series QS BA RN AP BF TS MQ NP; median of the steady calls" % slices,
                       "blocks_per_gpu": r["blocks"], "plain_bytes_per_gpu": r["plain_bytes"], "ratio": round(r["comp_bytes"] / r["plain_bytes"], 4),
                       "decode_GBps": round(sum_u / dec_s / 1e9, 3), "on_disk_methods": r["methods"], "verified": True,
                       "format_parity": "rANS Nx16 / range coder / tok3 UNPINNED against htscodecs"},
            "roofline": {"bound": "hbm", "achieved": round(alg_enc / r["encode_s"] / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg_enc / r["encode_s"] / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                         "kernel": "whole call (several kernels + PCIe): hge::ransnx16_encode_kernel dominates (profiles/)",
                         "kernel_ms": round(r["encode_s"] * 1e3, 2), "algorithmic_bytes_per_launch": int(alg_enc),
                         "decode_achieved": round(alg_dec / r["decode_s"] / 1e9, 3)},
            "cpu_baseline": None if run.world != 1 or run.args.no_cpu_baseline else cpu_baseline_cram(r["sample"], max(1, run.ncores - 2), "enc"),
            "cpu_baseline_decode": None if run.world != 1 or run.args.no_cpu_baseline else cpu_baseline_cram(r["sample"], max(1, run.ncores - 2), "dec")}, ok

def bench_bam(args, eng, comp, desc, total_u, d_comp, d_desc, d_plain, d_status, dev, rank, world, seed):
    """frame every record of the inflated BAM (bam_read1's framing + checks) and decode all bases
    (nibble2base), stream resident in HBM.  A step = one framing pass + one base-decoding pass over the whole stream."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from htslib_amd import _native as nat, synth
    stream = torch.cuda.current_stream().cuda_stream
    nblocks = len(desc)
    eng.bgzf_inflate_dev(d_comp.data_ptr(), len(comp), d_desc.data_ptr(), nblocks, d_plain.data_ptr(), total_u, d_status.data_ptr(), stream)
    torch.cuda.synchronize()
    head = d_plain[:1 << 20].cpu().numpy().tobytes()
    n_ref, first = C.c_int32(), C.c_uint64()
    nat.check(nat.lib.hg_bam_header_host(head, len(head), C.byref(n_ref), C.byref(first)), "bam header")
    bad = C.c_uint64()
    n = nat.lib.hg_bam_frame_dev(eng._h, d_plain.data_ptr(), total_u, first.value, n_ref.value, None, 0, C.byref(bad), stream)
    assert n > 0, n
    d_off = torch.zeros(n, dtype=torch.int64, device=dev)
    d_boff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    tot = C.c_uint64()
    nat.check(nat.lib.hg_bam_bases_dev(eng._h, d_plain.data_ptr(), d_off.data_ptr(), 0, d_boff.data_ptr(), None, 0, C.byref(tot), stream), "bases")
    d_bases = torch.empty(int(n) * 160 + 4096, dtype=torch.uint8, device=dev)

    def step():
        m = nat.lib.hg_bam_frame_dev(eng._h, d_plain.data_ptr(), total_u, first.value, n_ref.value, d_off.data_ptr(), n, C.byref(bad), stream)
        assert m == n
        nat.check(nat.lib.hg_bam_bases_dev(eng._h, d_plain.data_ptr(), d_off.data_ptr(), n, d_boff.data_ptr(), d_bases.data_ptr(),
                                           d_bases.numel(), C.byref(tot), stream), "bases")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # `samtools index` on the same resident stream (one call, reported beside the headline of this op)
    import struct
    nr = n_ref.value
    pq = 8 + struct.unpack_from("<I", head, 4)[0] + 4
    rl = []
    for _ in range(nr):
        ln = struct.unpack_from("<i", head, pq)[0]
        rl.append(struct.unpack_from("<I", head, pq + 4 + ln)[0]); pq += 8 + ln
    rl = np.array(rl + [0], dtype=np.uint32)
    bai_out = C.create_string_buffer(256 << 20)
    dsc = np.ascontiguousarray(desc)
    # (the bench stream is a concatenation of independently sorted 32 MiB chunks; an index exists for one sorted chunk)
    n_idx = min(int(n), 100_000)
    len_idx = int(d_off[n_idx].item()) if n_idx < n else int(total_u)
    t0b = time.perf_counter()
    bai_len = nat.lib.hg_bai_build_dev(eng._h, d_plain.data_ptr(), len_idx, first.value, nr, rl.ctypes.data, d_off.data_ptr(), n_idx,
                                       dsc.ctypes.data, len(dsc), len(comp), bai_out, len(bai_out), stream)
    bai_ms = (time.perf_counter() - t0b) * 1e3
    # verification against the oracle on the first chunk (outside the timed region)
    ok = True
    try:
        from tests.test_bam_frame import BamOracle
        orc = BamOracle()
        chk = d_plain[:8 << 20].cpu().numpy().tobytes()
        off = d_off[:20000].cpu().numpy().astype(np.uint64)
        lim = min(int((off < (8 << 20) - 70000).sum()), len(off) - 1)
        wn, _, woff = orc.frame(chk[:int(off[lim])], first.value)
        ok = wn == lim and bool((woff == off[:lim]).all())
        boff = d_boff[:lim + 1].cpu().numpy(); bases = d_bases[:int(boff[lim])].cpu().numpy().tobytes()
        ok = ok and all(bases[boff[i]:boff[i + 1]] == orc.bases(chk, int(off[i])) for i in range(0, lim, 97))
        # CPU baseline: the oracle's scalar port of the same two loops, one core, on the sample
        cpu = None
        if rank == 0 and not args.no_cpu_baseline:
            samp = chk[:int(off[lim])]
            t = time.perf_counter()
            for _ in range(3):
                wn2, _, woff2 = orc.frame(samp, first.value)
                buf = C.create_string_buffer(1 << 16)
                for o in woff2:
                    x = int(o) + 4
                    lq, nc, ls = samp[x + 8], int.from_bytes(samp[x + 12:x + 14], "little"), int.from_bytes(samp[x + 16:x + 20], "little")
                    orc.L.orc_nibble2base(samp[x + 32 + lq + 4 * nc:x + 32 + lq + 4 * nc + (ls + 1) // 2], buf, ls)
            dt = (time.perf_counter() - t) / 3
            cpu = {"value": round(len(samp) / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                   "sample": "oracle/bam_oracle.c framing + nibble2base driven record by record from Python over the first %.1f MB "
                             "(call overhead included; the reference's bam_read1 cannot be built without all of libhts)" % (len(samp) / 1e6)}
    except Exception as e:  # oracle not built on this box
        ok = ok and False
        cpu = None
        print("verification unavailable:", e, file=sys.stderr)
    from htslib_amd.bgzf import reduce_timing
    elapsed, sum_u, sum_n, ok = reduce_timing(elapsed, float(total_u), float(n), ok, world, dev)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        alg = total_u / 290.0 * 36 + tot.value / 2 + tot.value        # core fields read + packed bases read + ASCII written, per rank
        return ({"metric": "BAM record framing + base decoding throughput, uncompressed BAM GB/s (HBM-resident)",
                          "value": round(sum_u * args.steps / elapsed / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                          "config": {"workload": "bam_read1 framing + nibble2base over a %.1f GiB inflated synthetic BAM per GPU" % (total_u / 2**30),
                                     "records_per_gpu": int(n), "records_per_s": round(sum_n * args.steps / elapsed, 1), "verified": bool(ok),
                                     "bai_build": {"records": n_idx, "ms": round(bai_ms, 2), "bai_bytes_or_error": int(bai_len)}},
                          "roofline": {"bound": "hbm", "achieved": round(alg / (ms / 1e3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(alg / (ms / 1e3) / 1e9 / 8000.0, 4), "traffic": None,
                                       "algorithmic_bytes_per_launch": int(alg)},
                          **({"cpu_baseline": cpu} if cpu else {})}, ok)
    return None, ok



def op_deflate(run: Run, S: Staged, steps: int, warmup: int):
    """BASELINE configs[2]: BGZF deflate of the same BAM; the plain image is produced on the device by the inflate kernel,
    re-cut into the same blocks, compressed, then verified by inflating the GPU-written stream again (and, for a slice of
    it, with the real reference)."""
    import torch
    from htslib_amd import _native as nat
    from htslib_amd.bgzf import reduce_timing
    args = run.args
    S.eng.bgzf_inflate_dev(S.d_comp.data_ptr(), S.comp_len, S.d_desc.data_ptr(), S.nblocks, S.d_out.data_ptr(), S.total_u,
                           S.d_status.data_ptr(), S.stream)
    torch.cuda.synchronize()
    assert int((S.d_status != 0).sum()) == 0
    nblocks, total_u, d_plain = S.nblocks, S.total_u, S.d_out
    ddesc = S.desc.copy()
    ddesc["coff"] = np.arange(nblocks, dtype=np.uint64) * 65536
    d_ddesc = torch.from_numpy(ddesc.view(np.uint8).reshape(-1).copy()).to(run.dev)
    d_slots = torch.empty(nblocks * 65536 + 256, dtype=torch.uint8, device=run.dev)
    d_clen = torch.zeros(nblocks, dtype=torch.int32, device=run.dev)

    def step():
        nat.check(nat.lib.hg_bgzf_deflate_dev(S.eng._h, d_plain.data_ptr(), d_ddesc.data_ptr(), nblocks, args.level,
                                              d_slots.data_ptr(), d_clen.data_ptr(), S.stream), "deflate")

    elapsed, k_ms = run.timed(step, steps, warmup)
    # verify: pack, inflate the GPU-written stream on the GPU, compare with the plain image
    d_packed = torch.empty(nblocks * 65536 + 256, dtype=torch.uint8, device=run.dev)
    d_poff = torch.zeros(nblocks + 1, dtype=torch.int64, device=run.dev)
    d_total = torch.zeros(1, dtype=torch.int64, device=run.dev)
    nat.check(nat.lib.hg_bgzf_pack_dev(S.eng._h, d_slots.data_ptr(), d_ddesc.data_ptr(), d_clen.data_ptr(), nblocks,
                                       d_packed.data_ptr(), d_packed.numel(), d_poff.data_ptr(), d_total.data_ptr(), 0,
                                       S.stream), "pack")
    torch.cuda.synchronize()
    del d_slots
    comp_len = int(d_total.item())
    rdesc = S.desc.copy()
    rdesc["coff"] = d_poff[:nblocks].cpu().numpy().astype(np.uint64)
    rdesc["clen"] = d_clen.cpu().numpy().astype(np.uint32)
    d_rdesc = torch.from_numpy(rdesc.view(np.uint8).reshape(-1).copy()).to(run.dev)
    d_back = torch.empty(total_u + 256, dtype=torch.uint8, device=run.dev)
    st2 = torch.full((nblocks,), 77, dtype=torch.int32, device=run.dev)
    S.eng.bgzf_inflate_dev(d_packed.data_ptr(), comp_len, d_rdesc.data_ptr(), nblocks, d_back.data_ptr(), total_u,
                           st2.data_ptr(), S.stream)
    torch.cuda.synchronize()
    ok = int((st2 != 0).sum()) == 0 and bool(torch.equal(d_back[:total_u], d_plain[:total_u]))
    del d_back
    ref_ok = None
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bgzip_ld")
    if run.rank == 0 and os.path.exists(exe):                              # the REAL reference decodes a slice of our stream
        cut = min(nblocks, 2000)
        end = int(rdesc["coff"][cut - 1] + rdesc["clen"][cut - 1])
        blob = d_packed[:end].cpu().numpy().tobytes()
        want = d_plain[:int(S.desc["uoff"][cut - 1] + S.desc["ulen"][cut - 1])].cpu().numpy().tobytes()
        p = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"htsgpu_defl_{os.getpid()}.gz")
        open(p, "wb").write(blob)
        r = subprocess.run([exe, "-d", "-c", p], capture_output=True)
        os.unlink(p)
        ref_ok = r.returncode == 0 and r.stdout == want
        ok = ok and ref_ok
    del d_packed
    elapsed, sum_u, sum_c, ok = reduce_timing(elapsed, float(total_u), float(comp_len), ok, run.world, run.reduce_dev)
    if run.rank != 0:
        return None, ok
    alg = float(total_u + comp_len)
    out = {"metric": "BGZF deflate throughput, uncompressed GB/s (encode, HBM-resident)", "value": round(sum_u * steps / elapsed / 1e9, 3),
           "unit": "GB/s", "n_gpus": run.world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed * 1e3 / steps, 3), "higher_is_better": True, "scaling": "strong" if S.shard is not None else "weak",
           "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": (f"BGZF deflate (level {args.level}) of ONE {args.gib:g} GiB synthetic BAM, block ranges split over the ranks (shard_blocks), "
                                   if S.shard is not None else f"BGZF deflate (level {args.level}) of a {args.gib:g} GiB synthetic BAM per GPU, ") +
                                  "blocks cut as bam_write1/bgzf_flush_try would", "blocks_per_gpu": nblocks,
                      "plain_bytes_per_gpu": int(total_u), "compressed_bytes_per_gpu": int(comp_len),
                      "ratio": round(total_u / comp_len, 3), "zlib6_ratio": round(total_u / S.comp_len, 3),
                      "size_vs_zlib6": round(comp_len / S.comp_len, 4), "verified": bool(ok),
                      "decodes_with_reference_htslib": ref_ok},
           "roofline": {"bound": "hbm", "achieved": round(alg / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic": hbm_traffic_file("hbm_traffic_deflate.json", ["fetch_bytes_per_plain_byte", "write_bytes_per_plain_byte"], total_u),
                        "kernel": "hgd::bgzf_deflate_kernel", "kernel_ms": round(k_ms, 3),
                        "algorithmic_bytes_per_launch": int(alg)}}
    if run.world == 1 and not args.no_cpu_baseline:
        lim = min(int(total_u), 1 << 30)
        cb = cpu_baseline_deflate(d_plain[:lim].cpu().numpy().tobytes(), args.level, run.ncores)
        if cb:
            out["cpu_baseline"] = cb
    return out, ok


def op_e2e(run: Run, S: Staged):
    """What a libhts caller sees: a bgzf_read loop and a bgzf_write loop through htslib_amd/libhts_bgzf.so on a file in
    /dev/shm (positional reads into a pinned window -> kernel reading it over PCIe -> D2H -> caller's buffer, overlapped on 4 pipes), and the
    latency of 1000 random bgzf_seek + 100-byte reads.  The same three loops run against the REAL reference library
    (oracle/_ref/libref_bgzf_ld.so: bgzf.c + libdeflate with bgzf_mt(all cores)) on the same file."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    from tests import bgzf_capi
    from htslib_amd import synth
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(shm, f"htsgpu_e2e_{os.getpid()}_{run.rank}.bam")
    with open(path, "wb") as f:
        f.write(S.comp); f.write(synth.BGZF_EOF)
    res = {"file_GiB_plain": round(S.total_u / 2**30, 2)}
    chunk = 8 << 20
    rng = np.random.default_rng(12345)
    pick = rng.integers(0, S.nblocks, 1000)
    seeks = [(int(S.desc["coff"][i]) << 16) | int(rng.integers(0, max(1, int(S.desc["ulen"][i]) - 100))) for i in pick]

    def loops(L, threads, label):
        out = {}
        P = C.POINTER(bgzf_capi.BGZF)
        buf = C.create_string_buffer(chunk)
        # ---- sequential read
        fp = L.bgzf_open(path.encode(), b"r")
        if not fp:
            return None
        if threads:
            L.bgzf_mt(fp, threads, 256)
        t = time.perf_counter(); tot = 0
        while True:
            n = L.bgzf_read(fp, buf, chunk)
            if n <= 0:
                break
            tot += n
        dt = time.perf_counter() - t
        L.bgzf_close(fp)
        out["read_GBps"] = round(tot / dt / 1e9, 3) if tot == S.total_u else None
        if label == "gpu":
            # the same loop on a second handle: the first one left its device context and pinned windows parked for it
            fp = L.bgzf_open(path.encode(), b"r")
            t = time.perf_counter(); tot = 0
            while True:
                n = L.bgzf_read(fp, buf, chunk)
                if n <= 0:
                    break
                tot += n
            dt = time.perf_counter() - t
            L.bgzf_close(fp)
            out["read_next_handle_GBps"] = round(tot / dt / 1e9, 3) if tot == S.total_u else None
        # ---- random access
        fp = L.bgzf_open(path.encode(), b"r")
        if threads:
            L.bgzf_mt(fp, min(threads, 4), 256)
        small = C.create_string_buffer(128)
        t = time.perf_counter(); good = 0
        for vo in seeks:
            if L.bgzf_seek(fp, vo, 0) == 0 and L.bgzf_read(fp, small, 100) == 100:
                good += 1
        dt = time.perf_counter() - t
        L.bgzf_close(fp)
        out["seek_read_us"] = round(dt / len(seeks) * 1e6, 1) if good == len(seeks) else None
        return out

    # plain bytes for the write loop: one sequential read through OUR library into host memory (bounded to 4 GiB)
    ours = bgzf_capi.load()
    # the first handle of a process pays for the HIP runtime and the context (not a property of the data path): timed on its own
    t = time.perf_counter()
    fp0 = ours.bgzf_open(path.encode(), b"r")
    small0 = C.create_string_buffer(4096)
    ours.bgzf_read(fp0, small0, 4096)
    ours.bgzf_close(fp0)
    res["first_handle_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    res["gpu"] = loops(ours, 4, "gpu")
    lim = min(S.total_u, 4 << 30)
    host = np.empty(lim, dtype=np.uint8)
    fp = ours.bgzf_open(path.encode(), b"r"); got = 0
    while got < lim:
        n = ours.bgzf_read(fp, host.ctypes.data + got, min(chunk, lim - got))
        if n <= 0:
            break
        got += n
    ours.bgzf_close(fp)

    def write_loop(L, threads):
        wpath = path + ".w"
        fp = L.bgzf_open(wpath.encode(), b"w")
        if not fp:
            return None
        if threads:
            L.bgzf_mt(fp, threads, 256)
        t = time.perf_counter(); pos = 0
        while pos < got:
            n = min(chunk, got - pos)
            if L.bgzf_write(fp, C.cast(host.ctypes.data + pos, C.c_char_p), n) != n:
                return None
            pos += n
        rc = L.bgzf_close(fp)
        dt = time.perf_counter() - t
        sz = os.path.getsize(wpath)
        os.unlink(wpath)
        return {"write_GBps": round(got / dt / 1e9, 3) if rc == 0 else None, "out_bytes": sz}

    res["gpu"].update(write_loop(ours, 4) or {})
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libref_bgzf_ld.so")
    if os.path.exists(ref_so) and not run.args.no_cpu_baseline:
        R = C.CDLL(ref_so)
        P = C.POINTER(bgzf_capi.BGZF)
        R.bgzf_open.restype = P; R.bgzf_open.argtypes = [C.c_char_p, C.c_char_p]
        R.bgzf_close.argtypes = [P]; R.bgzf_mt.argtypes = [P, C.c_int, C.c_int]
        R.bgzf_read.restype = C.c_ssize_t; R.bgzf_read.argtypes = [P, C.c_void_p, C.c_size_t]
        R.bgzf_write.restype = C.c_ssize_t; R.bgzf_write.argtypes = [P, C.c_char_p, C.c_size_t]
        R.bgzf_seek.restype = C.c_int64; R.bgzf_seek.argtypes = [P, C.c_int64, C.c_int]
        nthr = min(run.ncores, 64)
        res["reference"] = loops(R, nthr, "ref")
        if res["reference"] is not None:
            res["reference"].update(write_loop(R, nthr) or {})
            res["reference"]["threads"] = nthr
    try:
        res["libhts_view"] = libhts_view(run, path, S.total_u)
    except Exception as e:                                                  # an auxiliary figure never takes the others down
        res["libhts_view"] = {"error": repr(e)}
    os.unlink(path)
    res["note"] = ("bgzf_read / bgzf_write loops with 8 MiB buffers on a /dev/shm file; gpu = htslib_amd/libhts_bgzf.so (4 pipes; first_handle_ms = open + 4 KiB read + close of the first handle of the process, i.e. HIP runtime + context creation, timed apart; "
                   "bgzf_mt called so the writer batches), reference = oracle/_ref/libref_bgzf_ld.so with bgzf_mt(threads); "
                   "write loop over the first %.1f GiB; seek_read_us = mean of 1000 random bgzf_seek + 100-byte bgzf_read" % (got / 2**30))
    return res


def libhts_view(run: Run, bam_path: str, plain_bytes: int, nreads: int = 12_000_000):
    """The drop-in INSIDE libhts, as samtools sees it: the reference's own test/test_view.c (a small `samtools view`) linked to
    oracle/_ref/libhts_gpu.so (the reference's libhts objects minus bgzf.o, on our bgzf_front.cpp / cram_block_front.cpp) against the same program on
    the reference's whole libhts (oracle/_ref/ref_view, libdeflate flavour, -@threads), both on the synthetic BAM in /dev/shm:
      decode  = view -@T -B -N n in.bam           (bam_read1 loop over bgzf_read, nothing written -- `samtools view -c`)
      bam2bam = view -@T -b -N n -p out.bam in.bam (bam_read1 -> bam_write1 -> bgzf_write at level 6 -- `samtools view -b`)
    Wall clock of the whole process (for ours that includes loading the HIP runtime), first n records (about n x 309 plain bytes)."""
    import subprocess
    gpu, ref = os.path.join(ROOT, "oracle", "_ref", "ref_view_gpu"), REF_VIEW
    if not (os.path.exists(gpu) and os.path.exists(ref)):
        return None
    out = bam_path + ".view.bam"
    plain = min(plain_bytes, int(nreads * 308.6))
    nthr = min(run.ncores, 64)

    def one(exe, threads, mode):
        cmd = [exe, "-@", str(threads), "-N", str(nreads)] + (["-B"] if mode == "decode" else ["-b", "-p", out]) + [bam_path]
        t = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
        dt = time.perf_counter() - t
        sz = os.path.getsize(out) if mode != "decode" and os.path.exists(out) else None
        if sz is not None: os.unlink(out)
        if p.returncode != 0:
            return {"error": p.stderr.decode("latin1")[-300:]}
        r = {"seconds": round(dt, 3), "plain_GBps": round(plain / dt / 1e9, 3), "M_records_per_s": round(nreads / dt / 1e6, 3), "threads": threads}
        if sz is not None: r["out_bytes"] = sz
        return r

    res = {"records": nreads, "plain_GB": round(plain / 1e9, 3)}
    ref_threads = sorted({t for t in (8, 16, nthr) if t <= max(8, nthr)})       # (-@4 never was the reference's best: 11 s for BAM -> BAM)
    for mode in ("decode", "bam2bam"):
        res[mode] = {"libhts_gpu": one(gpu, 4, mode)}
        if not run.args.no_cpu_baseline:
            # the reference at ITS best thread count (whole-process wall clock; more threads than the one bam_read1 / bam_write1 thread can feed only cost start-up)
            tries = [one(ref, t, mode) for t in ref_threads]
            good = [r for r in tries if "seconds" in r]
            res[mode]["reference"] = min(good, key=lambda r: r["seconds"]) if good else tries[-1]
            res[mode]["reference_by_threads"] = {str(r.get("threads", "?")): r.get("seconds") for r in tries}
    try:
        res.update(libhts_view_cram(run, gpu, ref, ref_threads))
    except Exception as e:
        res["cram_error"] = repr(e)
    res["note"] = ("the reference's test/test_view.c, unmodified, on oracle/_ref/libhts_gpu.so (reference libhts objects minus bgzf.o + our front-end, -@4 = bgzf_mt "
                   "-> device batches) vs on the reference's own libhts (libdeflate; best of -@%s); whole-process wall clock on a /dev/shm BAM; both are bounded by "
                   "the ONE thread that runs bam_read1 / bam_write1 (sam.c:784-928), which is why N1 (record framing on the device) exists; cram_decode / cram_encode: the same "
                   "two programs on a CRAM 3.0 file (default level: gzip + rANS 4x8 blocks) -- the reference's cram_decode_slice / cram_encode_slice on worker threads, "
                   "our cram_uncompress_block / cram_compress_block underneath" % ",".join(map(str, ref_threads)))
    return res


def libhts_view_cram(run: Run, gpu: str, ref: str, ref_threads, copies: int = 64, nrec: int = 10000):
    """libhts-level CRAM figures (north_star: "samtools/bcftools see a drop-in libhts"): test_view on libhts_gpu.so vs on the reference's libhts,
      cram_decode       = view -@T -B in.cram               (ours: cram_get_bam_seq = the whole-slice reader, htslib_amd/csrc/cram_record_front.c: runs of containers
                                                              decoded on the device, block codecs + cram_decode_slice + cram_to_bam; stock: cram_decode_slice on the pool)
      cram_decode_blocks = the same with HTS_GPU_CRAM_SLICE=0 (the reference's cram_decode_slice on our per-block entry points: round 6's first form)
      cram_encode       = view -@T -C -o version=3.0 in.bam  (ours: cram_put_bam_seq = the whole-slice writer, runs of records through the device record encoder + block
                                                              auto-tuner; stock: bam_read1 + cram_encode_slice + cram_compress_block3 on the pool; to /dev/null)
      cram_encode_blocks = the same with HTS_GPU_CRAM_SLICE=0 (the reference's cram_encode_slice on our cram_compress_block2)
      cram31_decode / cram31_encode = the same on CRAM 3.1, the reference's default version (stock on oracle/'s scalar 3.1 codecs: a floor)
      cram_decode_large / cram_to_bam_large / cram_encode_large = view -B / view -b / view -C on a file of 4 x the slices (10 240 000 records): what a run of 1 024 slices buys, and `samtools view -b in.cram`
    on 256 slices (2 560 000 records) of the record baselines' workload, at the writer's default level (gzip + rANS 4x8).  Whole-process wall clock, best of 2.
    A run of the reader takes 0.2-0.4 s whatever it holds (one 1.5 MB quality stream through the 4-way rANS decoder is one chain on one lane group, ~150 ms), a
    process pays ~0.2-0.3 s of HIP start-up before it and ~0.15 s of teardown after: stock htslib is through 2.56 M records before our first record is out."""
    import subprocess
    import numpy as np
    from htslib_amd import _native as nat, synth_cram
    if not have_ref_view(): return {}
    eng = nat.Engine(run.local)
    base = [synth_cram.make_slice(np.random.default_rng(7 + i), nrec, 150) for i in range(4)]
    w = RefCramWorkload(eng, base, copies)
    try:
        cram = os.path.join(w.dir, "in_l5.cram")
        r = subprocess.run([ref, "-C", "-o", "version=3.0", "-t", w.fa, "-p", cram, w.bam], capture_output=True)
        if r.returncode != 0: return {"cram_error": r.stderr.decode("latin1")[-300:]}
        plain = len(w.bam_bytes)

        def one(exe, threads, mode, W=None, cram_=None, plain_=None, reps=2):
            W = W or w; cram_ = cram_ or cram; plain_ = plain_ or plain
            env = dict(os.environ, HTS_GPU_CRAM_SLICE="0") if mode in ("cram_decode_blocks", "cram_encode_blocks") else None
            if mode.startswith("cram31") and exe == ref: env = dict(os.environ, ORC_STUB_CODECS31="1")     # the reference's 3.1 codecs here: oracle/'s scalar restatements (htscodecs absent)
            cmd = ([exe, "-@", str(threads), "-B", "-i", "reference=" + W.fa, cram_] if mode in ("cram_decode", "cram_decode_blocks", "cram31_decode") else
                   [exe, "-@", str(threads), "-C", "-t", W.fa, "-p", "/dev/null", W.bam] if mode == "cram31_encode" else
                   [exe, "-@", str(threads), "-b", "-i", "reference=" + W.fa, "-p", os.path.join(W.dir, "out.bam"), cram_] if mode == "cram_to_bam" else
                   [exe, "-@", str(threads), "-C", "-o", "version=3.0", "-t", W.fa, "-p", "/dev/null", W.bam])
            best = None
            for _ in range(reps):
                t = time.perf_counter()
                p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600, env=env)
                dt = time.perf_counter() - t
                if p.returncode != 0: return {"error": p.stderr.decode("latin1")[-300:], "threads": threads}
                best = dt if best is None else min(best, dt)
            return {"seconds": round(best, 3), "bam_GBps": round(plain_ / best / 1e9, 3), "M_records_per_s": round(W.nrec / best / 1e6, 3), "threads": threads}

        def both(mode, gpu_threads, ref_thr=None, **kw):
            tries_g = [one(gpu, t, mode, **kw) for t in gpu_threads]
            good = [x for x in tries_g if "seconds" in x]
            e = {"libhts_gpu": min(good, key=lambda x: x["seconds"]) if good else tries_g[-1]}
            if not run.args.no_cpu_baseline and not mode.endswith("_blocks"):
                tries = [one(ref, t, mode, **kw) for t in (ref_thr or ref_threads)]
                good = [x for x in tries if "seconds" in x]
                e["reference"] = min(good, key=lambda x: x["seconds"]) if good else tries[-1]
                e["reference_by_threads"] = {str(x.get("threads", "?")): x.get("seconds") for x in tries}
            return e

        out = {"cram_records": w.nrec, "cram_file_bytes": os.path.getsize(cram), "cram_bam_GB": round(plain / 1e9, 3)}
        out["cram_decode"] = both("cram_decode", (4, 16))
        out["cram_decode_blocks"] = both("cram_decode_blocks", (64,), reps=1)
        out["cram_encode"] = both("cram_encode", (4,))
        out["cram_encode_blocks"] = both("cram_encode_blocks", (64,), reps=1)
        # CRAM 3.1, the reference's default output version (rANS Nx16 + tok3): our writer's file, read by both; written by both.  The stock side decodes / encodes with the
        # scalar restatements under oracle/ (ORC_STUB_CODECS31=1) -- real htscodecs has SIMD rANS, so its figures are a floor, not the reference's speed
        c31 = os.path.join(w.dir, "ours31.cram")
        r = subprocess.run([gpu, "-@", "4", "-C", "-t", w.fa, "-p", c31, w.bam], capture_output=True)
        if r.returncode == 0:
            out["cram31_decode"] = both("cram31_decode", (4,), (64,), cram_=c31, reps=1)
            out["cram31_encode"] = both("cram31_encode", (4,), (64,), reps=1)
        else: out["cram31_error"] = r.stderr.decode("latin1")[-200:]
        w.close()
        # the large file: four times the slices (one run of 256 + one of 768 slices in the reader)
        w = RefCramWorkload(eng, base, 4 * copies)
        big = os.path.join(w.dir, "in_l5.cram")
        r = subprocess.run([ref, "-@", "32", "-C", "-o", "version=3.0", "-t", w.fa, "-p", big, w.bam], capture_output=True)
        if r.returncode != 0: return dict(out, cram_large_error=r.stderr.decode("latin1")[-300:])
        out["cram_large_records"] = w.nrec
        kw = dict(W=w, cram_=big, plain_=len(w.bam_bytes), reps=1)          # (one run each: these take seconds, and the reference is near its best at 16 threads on every leg above)
        out["cram_decode_large"] = both("cram_decode", (4,), (16,), **kw)
        out["cram_to_bam_large"] = both("cram_to_bam", (4,), (16,), **kw)
        out["cram_encode_large"] = both("cram_encode", (4,), (16,), **kw)
        return out
    finally:
        w.close()


REF_VIEW = os.path.join(ROOT, "oracle", "_ref", "ref_view")


def have_ref_view() -> bool:
    return os.path.exists(REF_VIEW) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_hts.so"))


class RefCramWorkload:
    """Inputs for timing the REFERENCE's record layer (oracle/_ref/ref_view = the reference's test/test_view.c on the reference's whole libhts, built by
    oracle/Makefile with a stand-in for the absent htscodecs: a CRAM <= 3.0 tool).  From synthetic slices: a sorted BAM file with stored (level 0)
    BGZF blocks, the FASTA, and the CRAM 3.0 file ref_view itself writes from them with level 0 (RAW blocks) -- so that a timed run is
    cram_decode_slice + cram_to_bam (decode) or bam_read1 + cram_encode_slice + container framing (encode) and no block codec."""
    def __init__(self, eng, slices, copies):
        import tempfile
        from htslib_amd import synth, synth_cram
        base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        self.dir = tempfile.mkdtemp(prefix="htsgpu_refcram_", dir=base)
        bam, names, seqs, self.nrec = synth_cram.bam_from_slices(eng, slices, copies)
        self.bam_bytes, self.names, self.seqs = bam, names, seqs
        self.bam = os.path.join(self.dir, "in.bam"); self.fa = os.path.join(self.dir, "ref.fa"); self.cram = os.path.join(self.dir, "in_l0.cram")
        open(self.bam, "wb").write(synth.bgzf_compress(bam, level=0))
        fai = []
        with open(self.fa, "wb") as f:
            for nm, sq in zip(names, seqs):
                f.write(b">" + nm.encode() + b"\n"); off = f.tell()
                for i in range(0, len(sq), 60): f.write(sq[i:i + 60] + b"\n")
                fai.append("%s\t%d\t%d\t60\t61\n" % (nm, len(sq), off))
        open(self.fa + ".fai", "w").write("".join(fai))
        r = subprocess.run([REF_VIEW, "-C", "-l", "0", "-o", "version=3.0", "-t", self.fa, "-p", self.cram, self.bam], capture_output=True)
        if r.returncode != 0: raise RuntimeError("ref_view -C failed: " + r.stderr.decode("latin1")[-400:])

    def decode_cmd(self, threads): return [REF_VIEW, "-@", str(threads), "-B", "-i", "reference=" + self.fa, self.cram]
    def encode_cmd(self, threads): return [REF_VIEW, "-@", str(threads), "-C", "-l", "0", "-o", "version=3.0", "-t", self.fa, "-p", "/dev/null", self.bam]

    def close(self):
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)


def time_ref_view(cmd, procs: int, seconds: float, env=None):
    """`procs` copies of the command side by side, each started again and again for ~`seconds` -> completed runs per second, all copies together"""
    import threading
    done = [0] * procs; bad = []
    t_end = time.perf_counter() + seconds
    def work(i):
        while time.perf_counter() < t_end:
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            if r.returncode != 0: bad.append(r.stderr.decode("latin1")[-300:]); return
            done[i] += 1
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(procs)]
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    if bad or not sum(done): return None, (bad[:1] or ["no run finished"])[0]
    return sum(done) / el, None


def op_cram31(run: Run, steps: int, copies: int = 16, nrec: int = 10000, flag_sets=None):
    """BASELINE configs[4] as a FILE: "full CRAM 3.1 encode (rANS + name tokeniser + range coder)" of sorted 150 bp reads -- hg_bam_to_cram_host2 with
    HG_CRAM_WRITE_V31 (flags 1 = rANS Nx16 + tok3, htslib's default 3.1 profile; flags 3 adds HG_CRAM_WRITE_ARITH = the range-coder method sets and TOKA, which is
    the configuration BASELINE names): BAM header walk, record encoder on the device (cram_encode_slice), every series block through the auto-tuner with the 3.1
    method sets, container framing + CRCs.  HOST entry point: the BAM goes up and the file comes back inside the timed call.  copies = 312 gives the configured
    shape, 1 248 slices of 10 000 reads per GPU (100 M reads over 8 GPUs).  Beside it: the REFERENCE's own writer (ref_view -C -o version=3.1 [-o use_arith=1], its
    record layer + auto-tuner; the 3.1 codecs behind the htscodecs stand-in are oracle/'s scalar restatements -- htscodecs is absent) on a 64-slice sample of the
    same reads, as many processes as the host has cores for."""
    run.init_device()
    import ctypes as C
    from htslib_amd import _native as nat, synth_cram
    eng = nat.Engine(run.local)
    rng = np.random.default_rng(7)
    base = [synth_cram.make_slice(rng, nrec, 150, tags=True) for _ in range(4)]
    env_flags = os.environ.get("HG_BENCH_CRAM31_FLAGS")
    flag_sets = flag_sets or ([int(env_flags)] if env_flags else [1])
    bam, names, seqs, total_rec = synth_cram.bam_from_slices(eng, base, copies)

    class RefSeq(C.Structure):
        _fields_ = [("bases", C.c_void_p), ("len", C.c_uint64)]
    keep = [C.create_string_buffer(q, len(q)) for q in seqs]
    arr = (RefSeq * len(keep))(*[RefSeq(C.addressof(k), len(q)) for k, q in zip(keep, seqs)])
    out = np.zeros(len(bam) + (1 << 20), np.uint8); tot = C.c_uint64(); n = C.c_uint64()
    bb = C.create_string_buffer(bam, len(bam))
    results = []
    for flags in flag_sets:
        ts = []
        for _ in range(max(2, steps) + 1):
            t = time.perf_counter()
            rc = nat.lib.hg_bam_to_cram_host2(eng._h, C.cast(bb, C.c_void_p), len(bam), C.cast(arr, C.c_void_p), len(keep), nrec, 5, flags, out.ctypes.data, len(out), C.byref(tot), C.byref(n))
            ts.append(time.perf_counter() - t)
            assert rc == 0 and n.value == total_rec, (rc, n.value)
        t = sorted(ts[1:])[len(ts[1:]) // 2]
        cram = bytes(out[:tot.value])
        # verification outside the timed region: our own whole-file decoder gives the records back
        back = np.zeros(len(bam) + (1 << 22), np.uint8); bt = C.c_uint64(); bn = C.c_uint64()
        cb = C.create_string_buffer(cram, len(cram))
        rc = nat.lib.hg_cram_file_to_bam_host2(eng._h, C.cast(cb, C.c_void_p), len(cram), C.cast(arr, C.c_void_p), len(keep), back.ctypes.data, len(back), C.byref(bt), C.byref(bn), 0, None)
        verified = rc == 0 and bn.value == total_rec
        del back, cb
        results.append({"metric": "full CRAM 3.1 file encode: BAM -> CRAM 3.1 (record encoder + block auto-tuner with rANS Nx16 / tok3%s + framing), M records/s, host entry point incl. PCIe" % (" / range coder" if flags & 2 else ""),
               "value": round(total_rec / t / 1e6, 3), "unit": "M records/s", "n_gpus": 1, "steps": max(2, steps), "warmup": 1, "ms_per_step": round(t * 1e3, 2), "higher_is_better": True,
               "dtype": "u8", "data": "synthetic", "verified": bool(verified),
               "config": {"workload": "%d slices x %d records x 150 bp on %d references, tags; BAM %.2f GB -> CRAM 3.1 %.3f GB; method sets: rANS Nx16 + tok3%s" % (total_rec // nrec, nrec, len(keep), len(bam) / 1e9, len(cram) / 1e9, " + range coder (use_arith)" if flags & 2 else ""),
                          "write_flags": flags, "bam_GBps": round(len(bam) / t / 1e9, 3), "cram_ratio": round(len(cram) / len(bam), 4),
                          "parity": "tests/test_reference_cram.py + tests/test_libhts_gpu.py: the reference's reader (htscodecs stand-in on oracle/'s codecs: dialect unpinned) reads these files back to the input records"},
               "roofline": {"bound": "hbm", "achieved": round((2 * len(bam) + len(cram)) / t / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((2 * len(bam) + len(cram)) / t / 1e9 / HBM_PEAK_GBS, 6),
                            "traffic": None, "kernel": "whole host call (PCIe both ways, record encoder, codec launches of 2-4 auto-tuner rounds: profiles/r05_cram31_writer_rounds.txt)", "algorithmic_bytes": int(2 * len(bam) + len(cram))}})
    del out, bb
    if run.world == 1 and not run.args.no_cpu_baseline and have_ref_view():
        w = RefCramWorkload(eng, base, 16)
        try:
            threads = 4; procs = max(1, min(64, run.ncores // threads))
            env = dict(os.environ, ORC_STUB_CODECS31="1")
            for res, flags in zip(results, flag_sets):
                cmd = [REF_VIEW, "-@", str(threads), "-C", "-o", "version=3.1"] + (["-o", "use_arith=1"] if flags & 2 else []) + ["-t", w.fa, "-p", "/dev/null", w.bam]
                rate, err = time_ref_view(cmd, procs, 8.0, env=env)
                res["cpu_baseline"] = {"error": err} if rate is None else {
                    "value": round(rate * w.nrec / 1e6, 3), "unit": "M records/s", "cores": procs * threads, "kind": "reference",
                    "sample": "the reference's writer (ref_view -C -o version=3.1%s: bam_read1, cram_encode_slice, cram_compress_block3, framing) with ORC_STUB_CODECS31=1 = "
                              "oracle/'s scalar restatements behind the htscodecs stand-in (NOT htscodecs): %d processes x -@%d, each on a %d-record (64-slice) BAM of the same reads, for 8 s" % (" -o use_arith=1" if flags & 2 else "", procs, threads, w.nrec)}
        finally:
            w.close()
    res = results[0]
    if len(results) > 1: res["other_method_sets"] = results[1:]
    return res


def cpu_baseline_reference_records(eng, slices, ncores: int, mode: str, seconds: float = 10.0):
    """cpu_baseline kind "reference" for the record layer: the reference's own cram_decode_slice (cram/cram_decode.c:2346) + cram_to_bam, or bam_read1 +
    cram_encode_slice (cram/cram_encode.c:1096-1209) + container framing, through its own thread pool: ncores/4 processes x 4 pool threads."""
    if not have_ref_view(): return None
    threads = 4; procs = max(1, min(64, ncores // threads))
    w = RefCramWorkload(eng, slices, 4)
    try:
        cmd = w.decode_cmd(threads) if mode == "decode" else w.encode_cmd(threads)
        rate, err = time_ref_view(cmd, procs, seconds)
        if rate is None: return {"error": err}
        return {"value": round(rate * w.nrec / 1e6, 3), "unit": "M records/s", "cores": procs * threads, "kind": "reference",
                "sample": "reference htslib (oracle/_ref/ref_view = test/test_view.c on the reference's libhts; htscodecs stand-in, CRAM 3.0, level 0 so no block codec runs): "
                          "%d processes x -@%d, each %s %d records (%d slices of 10 000 on %d references) again and again for %.0f s"
                          % (procs, threads, "reading (cram_decode_slice + cram_to_bam, -B: no output)" if mode == "decode" else "writing BAM -> CRAM to /dev/null (bam_read1 + cram_encode_slice)",
                             w.nrec, w.nrec // 10000, 4 * len(slices), seconds)}
    finally:
        w.close()


def cpu_baseline_records(base_slices, procs: int, seconds: float = 12.0):
    """The record decoder's own source (cram_records_core.h, the serial chain) compiled for the CPU -- tests/native/cram_records_host.cpp, test
    infrastructure, "not reference code": htslib's cram_decode_slice cannot be built here (htscodecs absent) -- on `procs` processes, each
    decoding the same synthetic slices in a loop for ~`seconds`."""
    import tempfile
    src = os.path.join(ROOT, "tests", "native", "cram_records_host.cpp")
    so = os.path.join(tempfile.gettempdir(), f"libcram_records_host_{os.getpid()}.so")
    if subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src]).returncode != 0:
        return None
    code = (
        "import sys, time, ctypes as C, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from htslib_amd import synth_cram\n"
        "from tests import test_cram_records as T\n"
        "L = C.CDLL(%r)\n"
        "vp = C.c_void_p\n"
        "L.hgr_host_records_bound.argtypes = [C.c_size_t, vp, C.c_int, vp, vp, vp, vp]\n"
        "L.hgr_host_decode_records.argtypes = [C.c_size_t, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, vp]\n"
        "rng = np.random.default_rng(7)\n"
        "sl = [synth_cram.make_slice(rng, %d, 150) for _ in range(%d)]\n"
        "keep = []; arr = T._slice_array(sl, keep); n = len(sl)\n"
        "nrec, cc, nc, ac = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()\n"
        "assert L.hgr_host_records_bound(n, arr, 3, C.byref(nrec), C.byref(cc), C.byref(nc), C.byref(ac)) == 0\n"
        "R = nrec.value\n"
        "i32 = [np.zeros(R, np.int32) for _ in range(9)]; i64 = [np.zeros(R, np.int64) for _ in range(4)]; u64 = [np.zeros(R, np.uint64) for _ in range(2)]\n"
        "cig = np.zeros(cc.value, np.uint32); nam = np.zeros(nc.value, np.uint8); so_ = np.zeros(R, np.uint64); sq = np.zeros(R * 150 + 64, np.uint8); ql = np.zeros(R * 150 + 64, np.uint8)\n"
        "ao = np.zeros(R, np.uint64); al = np.zeros(R, np.int32); ax = np.zeros(ac.value, np.uint8)\n"
        "cols = T.Cols(*[a.ctypes.data for a in i32 + i64 + u64 + [cig, nam, so_, sq, ql, ao, al, ax]])\n"
        "ro = np.zeros(n + 1, np.uint64); st = np.zeros(n, np.int32)\n"
        "print('ready', flush=True); sys.stdin.readline()\n"
        "t0 = time.perf_counter(); done = 0\n"
        "while time.perf_counter() - t0 < %f:\n"
        "    assert L.hgr_host_decode_records(n, arr, 3, 1, R, len(cig), len(nam), len(sq), len(ax), C.byref(cols), ro.ctypes.data, st.ctypes.data) == 0\n"
        "    done += R\n"
        "print(done, time.perf_counter() - t0, flush=True)\n" % (ROOT, so, base_slices[0]["nrec"], min(2, len(base_slices)), seconds))
    ps = [subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(procs)]
    try:
        for q in ps:
            if q.stdout.readline().strip() != "ready": return None
        for q in ps: q.stdin.write("go\n"); q.stdin.flush()
        tot = 0.0
        for q in ps:
            d, t = q.stdout.readline().split(); tot += int(d) / float(t)
    except Exception:
        return None
    finally:
        for q in ps: q.kill()
        try: os.unlink(so)
        except OSError: pass
    return {"value": round(tot / 1e6, 3), "unit": "M records/s", "cores": procs, "kind": "port",
            "sample": "NOT reference code: this repository's serial record decoder (cram_records_core.h) compiled for the CPU, %d processes x the same "
                      "synthetic slices for %.0f s (cram_decode_slice of htslib is not buildable here: htscodecs absent)" % (procs, seconds)}


def op_records(run: Run, steps: int, slices: int = 256, nrec: int = 10000, cpu: bool = True):
    """CRAM slices -> uncompressed BAM records, everything on the device and DEVICE-RESIDENT: the decoded blocks of the slices are
    staged in HBM once (hg_cram_batch_stage), a step is one hg_cram_batch_decode_bam_dev (cram_decode_slice + cram_to_bam, BAM stream left in
    HBM).  Synthetic EXTERNAL-only slices as current htslib writes them (htslib_amd/synth_cram.py) -> the data-parallel passes."""
    run.init_device()
    import numpy as np
    from htslib_amd import _native as nat, synth_cram
    eng = nat.Engine(run.local)
    rng = np.random.default_rng(7)
    base = [synth_cram.make_slice(rng, nrec, 150) for _ in range(4)]
    sl = [base[i % 4] for i in range(slices)]
    keep = []
    arr = nat.cram_slice_array(sl, keep)
    bases = slices * nrec * 150 + 4096
    in_bytes = sum(sum(len(d) for _, d in s["blocks"]) + len(s["refs"][0][2]) for s in sl)
    h = eng.cram_batch_stage(arr, slices, 3, 1, bases)
    steps = max(5, steps)
    try:
        for _ in range(2):
            d, nb, nr, nf, st = eng.cram_batch_decode_bam(h, slices)
        assert (st == 0).all() and nr == slices * nrec, (st, nr)
        ts = []
        for _ in range(steps):
            t = time.perf_counter(); d, nb, nr, nf, st = eng.cram_batch_decode_bam(h, slices); ts.append(time.perf_counter() - t)
        bam = eng.cram_batch_read_bam(h, nb)
    finally:
        eng.cram_batch_free(h)
    t = sorted(ts)[len(ts) // 2]                                        # median; the call returns when the stream is in HBM (it synchronises)
    # verification outside the timed region: the host entry point through the chain kernel gives the same stream; first record's name
    os.environ["HG_CRAM_RECORDS_PATH"] = "chain"
    try:
        chk_n = min(slices, 16)
        bam_k, _, st_k = eng.cram_decode_bam(arr, chk_n, 3, 1, [], bases, chk_n * nrec * 420)
    finally:
        del os.environ["HG_CRAM_RECORDS_PATH"]
    verified = bool((st_k == 0).all() and bytes(bam_k) == bytes(bam[:len(bam_k)]) and bytes(bam[36:36 + 8]) == base[0]["truth"][0]["name"])
    alg = in_bytes + int(nb)
    out = {"metric": "CRAM record decoding: slices -> uncompressed BAM records (cram_decode_slice + cram_to_bam on the device), M records/s, device-resident",
           "value": round(slices * nrec / t / 1e6, 3), "unit": "M records/s", "n_gpus": 1, "steps": steps, "warmup": 2, "ms_per_step": round(t * 1e3, 3),
           "higher_is_better": True, "dtype": "u8", "data": "synthetic", "verified": verified,
           "config": {"workload": "%d slices x %d records x 150 bp, EXTERNAL-only encodings as htslib writes them; blocks staged in HBM, BAM left in HBM" % (slices, nrec),
                      "slices_decoded_by_the_data_parallel_passes": int(nf), "bam_GBps": round(nb / t / 1e9, 3), "in_bytes": int(in_bytes), "bam_bytes": int(nb),
                      "timing": "median wall time of %d synchronous calls (each ends with the stream in HBM; two small read-backs inside)" % steps,
                      "parity": "decoder pinned on the reference's 34 CRAM fixtures vs their SAM/BAM twins; passes == chain decoder (tests/test_cram_records_fast.py); "
                                "this run: first %d slices' stream == the chain kernel's" % chk_n},
           "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                        "kernel": "the whole step (about thirty launches; profiles/ has the per-kernel split)", "algorithmic_bytes": int(alg),
                        "note": "algorithmic bytes = decoded blocks + reference spans read once + BAM bytes written once"}}
    if cpu and not run.args.no_cpu_baseline:
        ref = None
        try: ref = cpu_baseline_reference_records(eng, base, os.cpu_count() or 1, "decode")
        except Exception as e: out["cpu_baseline_error"] = repr(e)
        # (the port of our own decoder is timed only when the reference's cannot be: the default run is long enough)
        if ref and "value" in ref: out["cpu_baseline"] = ref
        else: out["cpu_baseline"] = cpu_baseline_records(base, max(1, (os.cpu_count() or 1) - 2))
    return out


def cpu_baseline_encode(bam_sample: bytes, nrec: int, per_slice: int, ref: bytes, procs: int, seconds: float = 8.0):
    """The record ENCODER's own source (cram_encode_core.h / cram_encode_plan.h) compiled for the CPU -- tests/native/cram_records_host.cpp:
    hgr_host_encode_slices, test infrastructure, "not reference code" (htslib's cram_encode_slice needs htscodecs and the whole cram_fd machinery) --
    on `procs` processes, each encoding the same `nrec` BAM records in a loop for ~`seconds`."""
    import tempfile
    src = os.path.join(ROOT, "tests", "native", "cram_records_host.cpp")
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    so = os.path.join(tempfile.gettempdir(), f"libcram_records_host_enc_{os.getpid()}.so")
    fb, fr = os.path.join(tmp, f"htsgpu_encbase_{os.getpid()}.bam"), os.path.join(tmp, f"htsgpu_encbase_{os.getpid()}.ref")
    if subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src]).returncode != 0:
        return None
    open(fb, "wb").write(bam_sample); open(fr, "wb").write(ref)
    code = (
        "import sys, time, ctypes as C, numpy as np\n"
        "L = C.CDLL(%r)\n"
        "vp = C.c_void_p\n"
        "class RefSeq(C.Structure):\n"
        "    _fields_ = [('bases', vp), ('len', C.c_uint64)]\n"
        "bam = open(%r, 'rb').read(); ref = open(%r, 'rb').read()\n"
        "nrec, per = %d, %d; ns = (nrec + per - 1) // per\n"
        "rb = C.create_string_buffer(ref, len(ref)); ra = (RefSeq * 1)(RefSeq(C.addressof(rb), len(ref)))\n"
        "rgp = (C.c_char_p * 1)()\n"
        "out = np.zeros(len(bam) * 6 + 65536 * ns + 4096, np.uint8); off = np.zeros(ns + 2, np.uint64); st = np.full(ns + 1, 9, np.int32)\n"
        "L.hgr_host_encode_slices.restype = C.c_long\n"
        "L.hgr_host_encode_slices.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_uint32, vp, C.c_int, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp, C.c_size_t, vp]\n"
        "def once():\n"
        "    n = L.hgr_host_encode_slices(bam, len(bam), nrec, per, C.cast(ra, vp), 1, C.cast(rgp, vp), 0, 0, out.ctypes.data, len(out), off.ctypes.data, ns + 1, st.ctypes.data)\n"
        "    assert n == ns and (st[:ns] == 0).all(), (n, st[:ns])\n"
        "once()\n"
        "print('ready', flush=True); sys.stdin.readline()\n"
        "t0 = time.perf_counter(); done = 0\n"
        "while time.perf_counter() - t0 < %f:\n"
        "    once(); done += nrec\n"
        "print(done, time.perf_counter() - t0, flush=True)\n" % (so, fb, fr, nrec, per_slice, seconds))
    ps = [subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(procs)]
    try:
        for q in ps:
            if q.stdout.readline().strip() != "ready": return None
        for q in ps: q.stdin.write("go\n"); q.stdin.flush()
        tot = 0.0
        for q in ps:
            d, t = q.stdout.readline().split(); tot += int(d) / float(t)
    except Exception:
        return None
    finally:
        for q in ps: q.kill()
        for f in (so, fb, fr):
            try: os.unlink(f)
            except OSError: pass
    return {"value": round(tot / 1e6, 3), "unit": "M records/s", "cores": procs, "kind": "port",
            "sample": "NOT reference code: this repository's record encoder (cram_encode_core.h / cram_encode_plan.h, the kernels' source) compiled for the CPU and run "
                      "from plain loops, %d processes x the same %d BAM records (one slice) for %.0f s (cram_encode_slice of htslib is not buildable here: htscodecs absent)"
                      % (procs, nrec, seconds)}


def op_encode(run: Run, steps: int, slices: int = 64, nrec: int = 10000, cpu: bool = True):
    """SURVEY 8f N2, write side: BAM records -> CRAM slices (hg_cram_encode_slices_host = cram_encode_slice + process_one_read on the device: survey, counting
    walk, prefix sums, writing walk).  HOST entry point: the BAM goes up and the series blocks come back over PCIe inside the timed call.  Input = the
    BAM stream the record decoder produced from synthetic slices; verified by decoding the slices again to the same stream."""
    run.init_device()
    import ctypes as C
    import numpy as np
    from htslib_amd import _native as nat, synth_cram
    eng = nat.Engine(run.local)
    rng = np.random.default_rng(7)
    base = [synth_cram.make_slice(rng, nrec, 150)]                       # one slice replicated: every record aligns to the same reference
    sl = [base[0] for i in range(slices)]
    keep = []
    arr = nat.cram_slice_array(sl, keep)
    bases = slices * nrec * 150 + 4096
    bam, rec_off, st = eng.cram_decode_bam(arr, slices, 3, 1, [], bases, slices * nrec * 420)
    assert (st == 0).all()
    bam = bytes(bam); n = slices * nrec
    ref = base[0]["refs"][0][2]

    class RefSeq(C.Structure):
        _fields_ = [("bases", C.c_void_p), ("len", C.c_uint64)]
    rb = C.create_string_buffer(ref, len(ref)); ra = (RefSeq * 1)(RefSeq(C.addressof(rb), len(ref)))
    out = np.zeros(len(bam) * 2 + 65536 * slices, np.uint8); off = np.zeros(slices + 2, np.uint64); stt = np.zeros(slices + 1, np.int32); tot = C.c_uint64()
    bb = C.create_string_buffer(bam, len(bam))
    ts = []
    steps = max(5, steps)
    for _ in range(steps + 1):
        t = time.perf_counter()
        rc = nat.lib.hg_cram_encode_slices_host(eng._h, C.cast(bb, C.c_void_p), len(bam), n, nrec, C.cast(ra, C.c_void_p), 1, None, 0, 0, out.ctypes.data, len(out), off.ctypes.data, slices + 1,
                                                stt.ctypes.data, C.byref(tot))
        ts.append(time.perf_counter() - t)
        assert rc == 0, rc
    t = sorted(ts[1:])[len(ts[1:]) // 2]
    res = {"metric": "CRAM record encoding: BAM records -> slice series blocks (cram_encode_slice on the device), M records/s, host entry point incl. PCIe",
            "value": round(n / t / 1e6, 3), "unit": "M records/s", "n_gpus": 1, "steps": steps, "warmup": 1, "ms_per_step": round(t * 1e3, 2), "higher_is_better": True, "dtype": "u8",
            "data": "synthetic", "config": {"workload": "%d slices x %d records x 150 bp from a %.2f GB BAM stream; series blocks %.2f GB" % (slices, nrec, len(bam) / 1e9, tot.value / 1e9),
                                            "bam_GBps": round(len(bam) / t / 1e9, 3), "parity": "writer: any valid CRAM is correct; tests/test_cram_encode.py decodes its output back to the same records / BAM bytes"},
            "roofline": {"bound": "hbm", "achieved": round((len(bam) + tot.value) / t / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((len(bam) + tot.value) / t / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": None, "kernel": "whole host call (PCIe both ways + three kernels); HG_CRAM_RECORDS_TIMING-style split in profiles/", "algorithmic_bytes": int(len(bam) + tot.value)}}
    if cpu and run.world == 1 and not run.args.no_cpu_baseline:
        try:                                                               # frame the first slice's records (bam_read1: block_size + bytes)
            at = 0
            for _ in range(nrec):
                at += 4 + int.from_bytes(bam[at:at + 4], "little")
            rb = None
            try: rb = cpu_baseline_reference_records(eng, [synth_cram.make_slice(np.random.default_rng(70 + i), nrec, 150) for i in range(3)] + base, run.ncores, "encode")
            except Exception as e: res["cpu_baseline_error"] = repr(e)
            if rb and "value" in rb: res["cpu_baseline"] = rb
            else:
                cb = cpu_baseline_encode(bam[:at], nrec, nrec, ref, max(1, run.ncores - 2))
                if cb: res["cpu_baseline"] = cb
        except Exception as e:
            res["cpu_baseline_error"] = repr(e)
    return res


def op_fqz(run: Run, steps: int, streams: int = 512, nrec: int = 2000):
    """fqzcomp (CRAM method 7) decode + encode through the host entry points: quality blocks of nrec x 150 bp reads (streams = 1 248, nrec = 10 000: the quality
    blocks of BASELINE configs[4]'s 1 250 slices per GPU).  One adaptive chain per block: the throughput is the number of blocks in flight x the chain's rate."""
    run.init_device()
    import numpy as np
    from htslib_amd import _native as nat
    eng = nat.Engine(run.local)
    rng = np.random.default_rng(8)
    rl = 150
    quals, lens = [], []
    for _ in range(4):
        q = np.clip(38 + np.cumsum(rng.integers(-2, 3, nrec * rl)) % 12 - np.tile(np.arange(rl) // 12, nrec), 2, 41).astype(np.uint8)
        quals.append(q.tobytes()); lens.append(np.full(nrec, rl, np.uint32))
    datas = [quals[i % 4] for i in range(streams)]
    fixed = os.environ.get("HG_BENCH_FQZ_STRAT")                      # diagnosis: every block with one strategy (default: the four presets in turn)
    args = (datas, [lens[i % 4] for i in range(streams)], [None] * streams, [int(fixed) if fixed else i % 4 for i in range(streams)])
    te, td, enc = [], [], None
    steps = max(3, steps)
    for _ in range(steps + 1):
        t = time.perf_counter(); enc = eng.fqz_encode_host(*args); te.append(time.perf_counter() - t)
        blocks = [(7, e, len(d)) for e, d in zip(enc, datas)]
        t = time.perf_counter(); outs, st = eng.cram_uncompress_blocks(blocks); td.append(time.perf_counter() - t)
    assert (st == 0).all() and all(o == d for o, d in zip(outs, datas))
    nb = sum(map(len, datas)); nc = sum(map(len, enc))
    med = lambda v: sorted(v[1:])[len(v[1:]) // 2]
    out = {"metric": "fqzcomp (CRAM method 7) decode, plain GB/s through the host entry points (PCIe included)", "value": round(nb / med(td) / 1e9, 3), "unit": "GB/s",
           "n_gpus": 1, "steps": steps, "warmup": 1, "ms_per_step": round(med(td) * 1e3, 2), "higher_is_better": True, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "%d quality blocks of %d x %d bp; one adaptive chain per block; medians" % (streams, nrec, rl), "encode_GBps": round(nb / med(te) / 1e9, 3),
                      "ratio": round(nc / nb, 4), "verified_blocks": len(datas), "format_parity": "UNPINNED against htscodecs (oracle/fqzcomp_oracle.c)"},
           "roofline": {"bound": "hbm", "achieved": round((nb + nc) / med(td) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((nb + nc) / med(td) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                        "kernel": "whole decode call (hgq::fqz_decode_kernel: one dependent chain per block, ~110 dependent instructions per quality with the next models fetched ahead -- instruction latency, not bandwidth; DESIGN.md 4.10)",
                        "algorithmic_bytes": int(nb + nc)}}
    if not run.args.no_cpu_baseline:
        nproc = max(1, run.ncores - 2)
        with mp.get_context("fork").Pool(nproc) as pool:
            parts = pool.map(_fqz_cpu_worker, [([(enc[i], datas[i], nrec) for i in range(4)], 8.0)] * nproc)
        out["cpu_baseline"] = {"value": round(sum(d / t for d, t in parts) / 1e9, 3), "unit": "GB/s", "cores": nproc, "kind": "port",
                               "sample": "NOT reference code (htscodecs is absent): oracle/fqzcomp_oracle.c decoding four of the same blocks in a loop for 8 s on %d processes" % nproc}
    return out


def _fqz_cpu_worker(task):
    import ctypes as C
    blocks, seconds = task
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    sz = C.c_size_t
    orc.orc_fqz_decode.argtypes = [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(sz), C.c_void_p, sz, C.POINTER(sz)]
    cap = max(len(b[1]) for b in blocks) + 4096
    buf = C.create_string_buffer(cap); got = sz(0); lens = (C.c_uint32 * (max(b[2] for b in blocks) + 16))()
    done = 0
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        for comp, plain, nrec in blocks:
            nr = sz(0); orc.orc_fqz_decode(comp, len(comp), buf, cap, C.byref(got), lens, len(lens), C.byref(nr)); done += len(plain)
    return done, time.perf_counter() - t



def compact(o, depth=0):
    """The ONE stdout line must fit the driver's 8 KB tail with every op's headline figures in it: long prose (samples, notes, parity statements) is
    cut, nested tables (the variant matrix) are dropped; the complete object goes to stderr and to gpurun_out/bench_full.json."""
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if k in ("variants", "on_disk_methods", "note", "parity", "timing", "format_parity", "sample_detail", "traffic_note", "sharding", "prep_seconds", "op_seconds"): continue
            if depth >= 2 and k in ("metric", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "n_gpus", "warmup", "algorithmic_bytes_per_launch", "algorithmic_bytes",
                                    "cpu_baseline_port", "in_bytes", "bam_bytes", "blocks_per_gpu", "plain_bytes_per_gpu", "compressed_bytes_per_gpu", "peak", "streams_per_gpu", "verified_streams",
                                    "slices_decoded_by_the_data_parallel_passes", "verified_blocks"): continue
            if depth >= 3 and k in ("unit", "out_bytes") and ("frac" in o or k == "out_bytes"): continue      # (a roofline's unit is GB/s everywhere)
            if k == "libhts_view" and isinstance(v, dict):
                # the libhts-level figures, flat: {leg: {gpu_s, gpu_threads, ref_s, ref_threads}}
                flat = {}
                for leg in ("decode", "bam2bam", "cram_decode", "cram_decode_blocks", "cram_encode", "cram_encode_blocks", "cram31_decode", "cram31_encode", "cram_decode_large", "cram_to_bam_large", "cram_encode_large"):
                    e = v.get(leg)
                    if not isinstance(e, dict): continue
                    g, r = e.get("libhts_gpu") or {}, e.get("reference") or {}
                    if leg.startswith("cram"):                           # seconds only (the line has to fit the driver's tail)
                        flat[leg] = {"gpu_s": g.get("seconds")}
                        if r: flat[leg]["ref_s"] = r.get("seconds")          # (the reference's best thread count is in the full object)
                        if "error" in g or "error" in r: flat[leg]["error"] = str(g.get("error", r.get("error")))[:60]
                        continue
                    else:
                        flat[leg] = {"libhts_gpu": {"seconds": g.get("seconds"), "plain_GBps": g.get("plain_GBps", g.get("bam_GBps")), "threads": g.get("threads")},
                                     "reference": {"seconds": r.get("seconds"), "plain_GBps": r.get("plain_GBps", r.get("bam_GBps")), "threads": r.get("threads")}}
                    if "error" in g: flat[leg]["libhts_gpu"]["error"] = str(g["error"])[:60]
                    if "error" in r: flat[leg].setdefault("reference", {})["error"] = str(r["error"])[:60]
                if "error" in v: flat["error"] = str(v["error"])[:80]
                out[k] = flat
                continue
            if isinstance(v, str):
                lim = (150 if k == "workload" else 110 if k in ("metric", "sample") else 70) if depth < 2 else (90 if k == "workload" else 32 if k == "kernel" else 44)
                out[k] = v if len(v) <= lim else v[:lim - 1] + "~"
            elif isinstance(v, (dict, list)):
                if depth < 4: out[k] = compact(v, depth + 1)
            else: out[k] = v
        return out
    if isinstance(o, list): return [compact(v, depth + 1) for v in o[:8]]
    return o


def main():
    t_main = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=10.0, help="plain GiB of synthetic BAM per GPU")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--workers", type=int, default=0, help="host processes for workload preparation")
    ap.add_argument("--op", choices=["all", "inflate", "deflate", "rans", "bam", "cram", "e2e", "records", "fqz", "encode", "cram31"], default="all",
                    help="all (default) = inflate headline (BASELINE configs[1]) + `extra`: deflate (configs[2]), rans (configs[3]), "
                         "cram (configs[4] shape) and the end-to-end bgzf_read / bgzf_write figures, in ONE JSON line; "
                         "a single op prints that op's line alone; bam = SURVEY 8f N1 (record framing + nibble2base)")
    ap.add_argument("--nway", type=int, choices=[4, 32], default=32, help="--op rans: 32-way (default) or 4-way (rANS 4x16) streams")
    ap.add_argument("--slices", type=int, default=0, help="CRAM slices of 10 000 reads (rans: default 1000 = 10 M reads; cram: default 256)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="strong: ONE data set split over the ranks by shard_blocks (--op inflate / deflate / all: the headline and the deflate extra)")
    ap.add_argument("--extra-steps", type=int, default=5, help="timed steps of the `extra` ops in --op all")
    ap.add_argument("--no-cache", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="--op rans: skip the 16-variant host-API matrix (profiling runs)")
    args = ap.parse_args()
    run = Run(args)
    ok = True
    out = None
    if args.op == "cram31":
        out = op_cram31(run, args.steps, max(1, (args.slices or 64) // 4))
    elif args.op in ("records", "fqz", "encode"):
        out = op_records(run, args.steps, args.slices or 256) if args.op == "records" else op_encode(run, args.steps, args.slices or 64) if args.op == "encode" else op_fqz(run, args.steps, args.slices or 512, 10000 if (args.slices or 0) >= 1000 else 2000)
    elif args.op in ("rans", "cram"):
        if args.op == "rans":
            out, ok = op_rans(run, args.steps, args.warmup, args.slices or 1000, args.nway)
        else:
            out, ok = op_cram(run, args.steps, args.slices or 256)
    else:
        S = stage(run)
        if args.op == "bam":
            out, ok = bench_bam(args, S.eng, S.comp, S.desc, S.total_u, S.d_comp, S.d_desc, S.d_out, S.d_status, run.dev, run.rank, run.world, S.seed)
        elif args.op == "deflate":
            out, ok = op_deflate(run, S, args.steps, args.warmup)
        elif args.op == "e2e":
            S.eng.bgzf_inflate_dev(S.d_comp.data_ptr(), S.comp_len, S.d_desc.data_ptr(), S.nblocks, S.d_out.data_ptr(), S.total_u, S.d_status.data_ptr(), S.stream)
            out = {"metric": "end-to-end bgzf_read / bgzf_write through libhts_bgzf.so", "end_to_end": op_e2e(run, S)} if run.rank == 0 else None
        else:
            out, ok = op_inflate(run, S, args.steps, args.warmup)
            if args.op == "all" and args.scaling == "weak":
                extra = {}
                op_s = {"bgzf_inflate": round(time.perf_counter() - t_main, 1)}       # wall seconds per op of this run (the full object only; CPU baselines included)
                def lap(name, t0): op_s[name] = round(time.perf_counter() - t0, 1)
                es = max(1, args.extra_steps)
                t0 = time.perf_counter()
                d, ok2 = op_deflate(run, S, es, 1); ok = ok and ok2
                if d: extra["bgzf_deflate"] = d
                lap("bgzf_deflate", t0)
                if run.rank == 0 and run.world == 1:
                    t0 = time.perf_counter()
                    try:
                        extra["end_to_end"] = op_e2e(run, S)
                    except Exception as e:                                  # never lose the headline to an auxiliary figure
                        extra["end_to_end"] = {"error": repr(e)}
                    lap("end_to_end", t0)
                t0 = time.perf_counter()
                ld_comp, ld_desc = S.comp, S.desc
                del S
                import torch
                torch.cuda.empty_cache()
                if run.world == 1:
                    try:
                        extra["bgzf_inflate_libdeflate6"] = inflate_libdeflate(run, ld_comp, ld_desc, es, args.level)
                    except Exception as e:
                        extra["bgzf_inflate_libdeflate6"] = {"error": repr(e)}
                    torch.cuda.empty_cache()
                del ld_comp, ld_desc
                if run.world == 1:
                    try:
                        extra["bgzf_inflate_variant_B"] = inflate_variant_b(run, es)
                    except Exception as e:
                        extra["bgzf_inflate_variant_B"] = {"error": repr(e)}
                lap("inflate_variants", t0)
                t0 = time.perf_counter()
                d, ok2 = op_rans(run, es, 1, args.slices or 1000); ok = ok and ok2
                if d: extra["cram_rans_nx16_decode"] = d
                lap("cram_rans_nx16_decode", t0); t0 = time.perf_counter()
                d, ok2 = op_rans(run, es, 1, args.slices or 1000, 4); ok = ok and ok2
                if d: extra["cram_rans_4x16_decode"] = d
                lap("cram_rans_4x16_decode", t0); t0 = time.perf_counter()
                d, ok2 = op_cram(run, es, 256 if not args.slices else args.slices); ok = ok and ok2
                if d: extra["cram_slices"] = d
                lap("cram_slices", t0)
                if run.rank == 0 and run.world == 1:                         # the "next" rows of SURVEY 8f built this round: small, guarded probes
                    for key, fn in (("cram_records_to_bam", lambda: op_records(run, 5)), ("cram_records_encode", lambda: op_encode(run, 5)), ("cram31_file_encode", lambda: op_cram31(run, 2, 312, flag_sets=[3, 1])),
                                    ("cram_fqzcomp", lambda: op_fqz(run, 3, 1248, 10000))):
                        t0 = time.perf_counter()
                        try:
                            extra[key] = fn()
                        except Exception as e:
                            extra[key] = {"error": repr(e)}
                        lap(key, t0)
                if out is not None:
                    out["extra"] = extra
                    out["op_seconds"] = op_s
    if run.rank == 0 and out is not None:
        full = json.dumps(out)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w").write(full + "\n")
        except OSError:
            pass
        print(full, file=sys.stderr, flush=True)
        line = json.dumps(compact(out) if args.op == "all" else out)
        print(line, flush=True)
    run.finish()
    if not ok:
        sys.exit(2)


if __name__ == "__main__":
    main()
