#!/usr/bin/env python3
"""bench.py -- BGZF inflate of a synthetic BAM on MI355X (BASELINE.json configs[1]).

A "step" is ONE pass of the hot path over the whole workload: every BGZF block
of the (per-GPU) synthetic BAM is inflated + CRC-checked by one launch of the
gfx950 kernel, inputs and outputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gib G]

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
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 32 << 20          # plain bytes generated + deflated by one worker task
GEN_VERSION = "v1"
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def _prep_chunk(args):
    """Worker: synthesise chunk `idx` and deflate it the way stock htslib (zlib, level 6) does."""
    seed, idx, nbytes, level, cache_dir = args
    from htslib_amd import synth
    key = f"{GEN_VERSION}_{seed:x}_{idx}_{nbytes}_{level}"
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
    best = None
    try:
        for _ in range(3):
            t = time.perf_counter()
            with open(os.devnull, "wb") as dn:
                r = subprocess.run([exe, "-d", "-c", "-@", str(threads), path], stdout=dn, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t
            if r.returncode != 0:
                return None
            best = dt if best is None else min(best, dt)
    finally:
        os.unlink(path)
    return {"value": round(plain_len / best / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "reference",
            "sample": f"oracle/_ref/ref_bgzip_ld -d -@{threads} (htslib bgzf.c + libdeflate 1.8, hts_tpool) on "
                      f"{plain_len / 2**30:.2f} GiB of the same BAM from /dev/shm, best of 3"}


def hbm_traffic_rans(plain_bytes):
    """Same for the rANS Nx16 decode kernel (profiles/hbm_traffic_rans.json: FETCH_SIZE doubled as calibrated, + WRITE_SIZE)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_rans.json")))
        return int(t["traffic_bytes_per_plain_byte"] * plain_bytes)
    except Exception:
        return None


def hbm_traffic(plain_bytes):
    """HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic_inflate.json:
    FETCH_SIZE + WRITE_SIZE per plain byte at the 10 GiB config), scaled to this workload; None if absent."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_inflate.json")))
        return int((t["fetch_bytes_per_plain_byte"] + t["write_bytes_per_plain_byte"]) * plain_bytes)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=10.0, help="plain GiB of synthetic BAM per GPU")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--workers", type=int, default=0, help="host processes for workload preparation")
    ap.add_argument("--op", choices=["inflate", "deflate", "rans", "bam", "cram"], default="inflate",
                    help="inflate = BASELINE configs[1] (default, the headline); deflate = configs[2]; "
                         "rans = configs[3] (CRAM 3.1 rANS Nx16 decode of QS+BA series); bam = SURVEY 8f N1: record framing "
                         "(bam_read1) + nibble2base over the inflated stream, on the device; cram = configs[4] shape: whole CRAM 3.1 "
                         "slices through the cram_compress_block2 auto-tuner (rANS Nx16 + tok3 + range coder), host entry points")
    ap.add_argument("--slices", type=int, default=1000, help="--op rans: CRAM slices of 10 000 reads (1000 = 10 M reads)")
    ap.add_argument("--no-cache", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    ncores = os.cpu_count() or 1
    workers = args.workers or max(1, (ncores - 4) // max(1, world))

    if args.op == "rans":
        return bench_rans(args, rank, world, local, ncores)
    if args.op == "cram":
        return bench_cram(args, rank, world, local)
    # ---------------- workload preparation (host, not timed, before HIP init) --------------
    total_bytes = int(args.gib * (1 << 30))
    cache = None if args.no_cache else os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "htsgpu_bench_cache")
    t0 = time.perf_counter()
    first_chunk = 0
    if world > 1 and cache:
        seed = 0x5EED0001
        comp, first_chunk = prepare_shared(rank, world, seed, total_bytes, args.level, workers, cache, rotate=args.op != "bam")
    else:
        seed = 0x5EED0001 + 1000003 * rank
        comp = prepare(seed, total_bytes, args.level, workers, cache)
    t_prep = time.perf_counter() - t0

    import torch
    from htslib_amd import _native as nat

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    eng = nat.Engine(local)
    desc, total_u = nat.bgzf_scan(comp)
    nblocks = len(desc)
    comp_len = len(comp)
    pad = (-comp_len) % 256 + 256
    h_comp = torch.frombuffer(bytearray(comp), dtype=torch.uint8)
    d_comp = torch.zeros(comp_len + pad, dtype=torch.uint8, device=dev)
    d_comp[:comp_len].copy_(h_comp)
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(dev)
    d_out = torch.empty(total_u + 256, dtype=torch.uint8, device=dev)
    d_status = torch.full((nblocks,), 77, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream

    if args.op == "bam":
        return bench_bam(args, eng, comp, desc, total_u, d_comp, d_desc, d_out, d_status, dev, rank, world, seed)
    if args.op == "deflate":
        return bench_deflate(args, eng, comp, desc, total_u, d_comp, d_desc, d_out, d_status, dev, rank, world,
                             ncores, t_prep, seed)

    def step():
        eng.bgzf_inflate_dev(d_comp.data_ptr(), comp_len, d_desc.data_ptr(), nblocks, d_out.data_ptr(), total_u,
                             d_status.data_ptr(), stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in evs]

    # ---------------- verification (outside the timed region) ------------------------------
    st = d_status.cpu().numpy()
    nbad = int((st != 0).sum())
    # every block's CRC-32 (computed by the host writer on the ORIGINAL bytes) was re-checked in-kernel;
    # additionally compare the first chunk byte-for-byte with a regenerated plain image
    from htslib_amd import synth
    chk = min(CHUNK, total_bytes)
    plain0, _, _ = synth.bam_stream(chk, seed, first_chunk, first_chunk == 0)
    got0 = d_out[:len(plain0)].cpu().numpy().tobytes()
    bytes_ok = got0 == plain0
    ok = nbad == 0 and bytes_ok

    from htslib_amd.bgzf import reduce_timing
    elapsed, sum_u, sum_c, ok = reduce_timing(elapsed, float(total_u), float(comp_len), ok, world, dev)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = sum_u * args.steps / elapsed / 1e9
        k_ms = float(np.mean(kern_ms))
        alg_bytes = float(total_u + comp_len)           # per launch on this rank: C + U
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "BGZF inflate throughput, uncompressed GB/s (decode, CRC-checked, HBM-resident)",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BGZF inflate of a {args.gib:g} GiB synthetic coordinate-sorted 150bp BAM per GPU "
                                   f"(zlib level {args.level} blocks as written by stock htslib, <=65280 B each)",
                       "blocks_per_gpu": nblocks, "plain_bytes_per_gpu": int(total_u),
                       "compressed_bytes_per_gpu": int(comp_len), "ratio": round(total_u / comp_len, 3),
                       "sharding": "independent blocks, static split, no collective", "verified": bool(ok),
                       "prep_seconds": round(t_prep, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": hbm_traffic(total_u),
                         "kernel": "hg::bgzf_inflate_kernel", "kernel_ms": round(k_ms, 3),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "traffic_note": "FETCH_SIZE + WRITE_SIZE PMC passes at face value; FETCH_SIZE is not a byte count for this access mix "
                                         "(profiles/hbm_traffic_inflate.json: calibration)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample: at most 4 GiB plain of the same stream
            lim = 4 << 30
            if total_u > lim:
                cut = int(np.searchsorted(desc["uoff"], lim))
                sample = comp[:int(desc["coff"][cut])]
                plen = int(desc["uoff"][cut])
            else:
                sample, plen = comp, int(total_u)
            cb = cpu_baseline(sample, plen, ncores)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(2)


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


def bench_rans(args, rank, world, local, ncores):
    """BASELINE configs[3]: CRAM 3.1 rANS Nx16 decode (order-1 32-way QS + order-0 32-way BA) of
    --slices x 10 000 reads per GPU.  The streams are produced by the gfx950 ENCODER (the oracle is
    used only as cpu_baseline); the timed region is the decode launch, device resident."""
    import ctypes as C
    import torch
    from htslib_amd import _native as nat
    from htslib_amd.bgzf import reduce_timing
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(min(64, max(1, (ncores - 4) // max(1, world)))) as pool:
        chunks = pool.starmap(_rans_series, [(0x5EED0001 + 7_000_003 * rank + 1000 * i, 1) for i in range(args.slices)])
    series = [c[0] for c in chunks]
    t_prep = time.perf_counter() - t0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    eng = nat.Engine(local)
    plains, flags = [], []
    for qs, ba in series:
        plains += [qs, ba]; flags += [5, 4]                      # QS: order-1 X32, BA: order-0 X32
    streams = []
    for i in range(0, len(plains), 64):                           # encode on the GPU, in batches
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
    d_in = torch.frombuffer(blob, dtype=torch.uint8).to(dev)
    d_desc = torch.from_numpy(desc.view(np.uint8).reshape(-1).copy()).to(dev)
    total_u = int(out_len.astype(np.uint64).sum()); total_c = int(in_len.astype(np.uint64).sum())
    d_out = torch.zeros(int(desc["out_off"][-1]) + int(out_len[-1]) + 64, dtype=torch.uint8, device=dev)
    d_status = torch.full((n,), 77, dtype=torch.int32, device=dev)
    d_scratch = torch.zeros(int(words.sum()) + 64, dtype=torch.int32, device=dev)
    d_sel = torch.arange(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()

    def step():
        nat.check(nat.lib.hg_ransnx16_decode_dev(eng._h, d_in.data_ptr(), d_desc.data_ptr(), None, 0, d_sel.data_ptr(), n,
                                                 d_out.data_ptr(), d_status.data_ptr(), d_scratch.data_ptr(), stream), "rans decode")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(); step(); b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    ok = int((d_status != 0).sum()) == 0
    host_out = d_out.cpu().numpy()
    for i in (0, 1, n - 2, n - 1):
        ok = ok and host_out[int(desc["out_off"][i]):int(desc["out_off"][i]) + int(out_len[i])].tobytes() == plains[i]
    elapsed, sum_u, sum_c, ok = reduce_timing(elapsed, float(total_u), float(total_c), ok, world, dev)
    if rank == 0:
        alg = float(total_u + total_c)
        out = {"metric": "CRAM 3.1 rANS Nx16 decode throughput, uncompressed GB/s (HBM-resident)",
               "value": round(sum_u * args.steps / elapsed / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": f"rANS Nx16 decode of {args.slices} CRAM slices x 10 000 x 150 bp per GPU: QS (order-1, 32-way) "
                                      "+ BA (order-0, 32-way) data series; streams written by the gfx950 encoder; format parity "
                                      "with htscodecs UNPINNED", "streams_per_gpu": n, "plain_bytes_per_gpu": total_u,
                          "compressed_bytes_per_gpu": total_c, "verified": bool(ok), "prep_seconds": round(t_prep, 1)},
               "roofline": {"bound": "hbm", "achieved": round(alg / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": hbm_traffic_rans(total_u),
                            "kernel": "hgn::ransnx16_decode_kernel<32>", "kernel_ms": round(k_ms, 3),
                            "algorithmic_bytes_per_launch": int(alg)}}
        if world == 1 and not args.no_cpu_baseline:
            orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
            orc.orc_ransnx16_uncompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
            buf = C.create_string_buffer(1_500_000 + 64); got = C.c_size_t(0)
            k = min(n, 24); t = time.perf_counter(); done = 0
            for i in range(k):
                orc.orc_ransnx16_uncompress(streams[i], len(streams[i]), buf, 1_500_064, C.byref(got)); done += got.value
            dt = time.perf_counter() - t
            out["cpu_baseline"] = {"value": round(done / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle/ransnx16_oracle.c (scalar C restatement, NOT reference code: htscodecs absent) "
                                             f"decoding {k} of the same streams on one core"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(2)


def bench_cram(args, rank, world, local):
    """BASELINE configs[4] shape on the GPUs of one node: every rank encodes (and decodes back) its own --slices CRAM 3.1
    slices of 10 000 reads -- 8 data series each -- through hg_cram_compress_blocks_metrics_host, i.e. the reference's
    cram_compress_block2 loop with its method auto-tuner, then hg_cram_uncompress_blocks_host.  These are HOST entry
    points (the reference hands the codecs malloc'd blocks), so unlike the other ops the rate includes PCIe both ways."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_cram_slices
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()
    S = args.slices if args.slices != 1000 else 256
    r = bench_cram_slices.main(S, device=local, reps=max(2, args.steps), quiet=True)
    from htslib_amd.bgzf import reduce_timing
    enc_s, sum_u, sum_c, ok = reduce_timing(r["encode_s"], float(r["plain_bytes"]), float(r["comp_bytes"]), True, world, dev)
    dec_s, _, _, _ = reduce_timing(r["decode_s"], float(r["plain_bytes"]), float(r["comp_bytes"]), True, world, dev)
    if rank == 0:
        print(json.dumps({"metric": "CRAM 3.1 slice encode throughput through the block-method auto-tuner, plain GB/s (host entry points, PCIe included)",
                          "value": round(sum_u / enc_s / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": max(2, args.steps), "warmup": 1,
                          "ms_per_step": round(enc_s * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u8", "data": "synthetic",
                          "config": {"workload": "%d slices x 10 000 reads per GPU, series QS BA RN AP BF TS MQ NP; best steady call" % S,
                                     "blocks_per_gpu": r["blocks"], "plain_bytes_per_gpu": r["plain_bytes"], "ratio": round(r["comp_bytes"] / r["plain_bytes"], 4),
                                     "decode_GBps": round(sum_u / dec_s / 1e9, 3), "on_disk_methods": r["methods"], "verified": True,
                                     "format_parity": "rANS Nx16 / range coder / tok3 UNPINNED against htscodecs"}}))
    if world > 1:
        dist.destroy_process_group()
    return 0


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
        print(json.dumps({"metric": "BAM record framing + base decoding throughput, uncompressed BAM GB/s (HBM-resident)",
                          "value": round(sum_u * args.steps / elapsed / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                          "config": {"workload": "bam_read1 framing + nibble2base over a %.1f GiB inflated synthetic BAM per GPU" % (total_u / 2**30),
                                     "records_per_gpu": int(n), "records_per_s": round(sum_n * args.steps / elapsed, 1), "verified": bool(ok),
                                     "bai_build": {"records": n_idx, "ms": round(bai_ms, 2), "bai_bytes_or_error": int(bai_len)}},
                          "roofline": {"bound": "hbm", "achieved": round(alg / (ms / 1e3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(alg / (ms / 1e3) / 1e9 / 8000.0, 4), "traffic": None,
                                       "algorithmic_bytes_per_launch": int(alg)},
                          **({"cpu_baseline": cpu} if cpu else {})}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def bench_deflate(args, eng, comp, desc, total_u, d_comp, d_desc, d_plain, d_status, dev, rank, world, ncores,
                  t_prep, seed):
    """BASELINE configs[2]: BGZF deflate level 6 of the same BAM; the plain image is produced on the
    device by the inflate kernel, re-cut into the same blocks, compressed, then verified by inflating
    the GPU-written stream again and comparing CRCs/bytes."""
    import torch
    import torch.distributed as dist
    from htslib_amd import _native as nat
    stream = torch.cuda.current_stream().cuda_stream
    nblocks = len(desc)
    eng.bgzf_inflate_dev(d_comp.data_ptr(), len(comp), d_desc.data_ptr(), nblocks, d_plain.data_ptr(), total_u,
                         d_status.data_ptr(), stream)
    torch.cuda.synchronize()
    assert int((d_status != 0).sum()) == 0
    ddesc = desc.copy()
    ddesc["coff"] = np.arange(nblocks, dtype=np.uint64) * 65536
    d_ddesc = torch.from_numpy(ddesc.view(np.uint8).reshape(-1).copy()).to(dev)
    d_slots = torch.empty(nblocks * 65536 + 256, dtype=torch.uint8, device=dev)
    d_clen = torch.zeros(nblocks, dtype=torch.int32, device=dev)

    def step():
        nat.check(nat.lib.hg_bgzf_deflate_dev(eng._h, d_plain.data_ptr(), d_ddesc.data_ptr(), nblocks, args.level,
                                              d_slots.data_ptr(), d_clen.data_ptr(), stream), "deflate")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(); step(); b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    # verify: pack, inflate the GPU-written stream on the GPU, compare with the plain image
    d_packed = torch.empty(nblocks * 65536 + 256, dtype=torch.uint8, device=dev)
    d_poff = torch.zeros(nblocks + 1, dtype=torch.int64, device=dev)
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    nat.check(nat.lib.hg_bgzf_pack_dev(eng._h, d_slots.data_ptr(), d_ddesc.data_ptr(), d_clen.data_ptr(), nblocks,
                                       d_packed.data_ptr(), d_packed.numel(), d_poff.data_ptr(), d_total.data_ptr(), 0,
                                       stream), "pack")
    torch.cuda.synchronize()
    comp_len = int(d_total.item())
    rdesc = desc.copy()
    rdesc["coff"] = d_poff[:nblocks].cpu().numpy().astype(np.uint64)
    rdesc["clen"] = d_clen.cpu().numpy().astype(np.uint32)
    d_rdesc = torch.from_numpy(rdesc.view(np.uint8).reshape(-1).copy()).to(dev)
    d_back = torch.empty(total_u + 256, dtype=torch.uint8, device=dev)
    st2 = torch.full((nblocks,), 77, dtype=torch.int32, device=dev)
    eng.bgzf_inflate_dev(d_packed.data_ptr(), comp_len, d_rdesc.data_ptr(), nblocks, d_back.data_ptr(), total_u,
                         st2.data_ptr(), stream)
    torch.cuda.synchronize()
    ok = int((st2 != 0).sum()) == 0 and bool(torch.equal(d_back[:total_u], d_plain[:total_u]))
    # a slice of the GPU-written stream must also decode with the REAL reference when it is present
    ref_ok = None
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bgzip_ld")
    if rank == 0 and os.path.exists(exe):
        cut = min(nblocks, 2000)
        end = int(rdesc["coff"][cut - 1] + rdesc["clen"][cut - 1])
        blob = d_packed[:end].cpu().numpy().tobytes()
        want = d_plain[:int(desc["uoff"][cut - 1] + desc["ulen"][cut - 1])].cpu().numpy().tobytes()
        p = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"htsgpu_defl_{os.getpid()}.gz")
        open(p, "wb").write(blob)
        r = subprocess.run([exe, "-d", "-c", p], capture_output=True)
        os.unlink(p)
        ref_ok = r.returncode == 0 and r.stdout == want
        ok = ok and ref_ok
    from htslib_amd.bgzf import reduce_timing
    elapsed, sum_u, sum_c, ok = reduce_timing(elapsed, float(total_u), float(comp_len), ok, world, dev)
    if rank == 0:
        value = sum_u * args.steps / elapsed / 1e9
        alg = float(total_u + comp_len)
        out = {"metric": "BGZF deflate throughput, uncompressed GB/s (encode, HBM-resident)", "value": round(value, 3),
               "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": f"BGZF deflate (level {args.level}) of a {args.gib:g} GiB synthetic BAM per GPU, "
                                      "blocks cut as bam_write1/bgzf_flush_try would", "blocks_per_gpu": nblocks,
                          "plain_bytes_per_gpu": int(total_u), "compressed_bytes_per_gpu": int(comp_len),
                          "ratio": round(total_u / comp_len, 3), "zlib6_ratio": round(total_u / len(comp), 3),
                          "size_vs_zlib6": round(comp_len / len(comp), 4), "verified": bool(ok),
                          "decodes_with_reference_htslib": ref_ok},
               "roofline": {"bound": "hbm", "achieved": round(alg / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                            "kernel": "hgd::bgzf_deflate_kernel", "kernel_ms": round(k_ms, 3),
                            "algorithmic_bytes_per_launch": int(alg)}}
        if world == 1 and not args.no_cpu_baseline and os.path.exists(exe):
            lim = min(int(total_u), 1 << 30)
            sample = d_plain[:lim].cpu().numpy().tobytes()
            p = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"htsgpu_defl_in_{os.getpid()}")
            open(p, "wb").write(sample)
            best = None
            for _ in range(2):
                t = time.perf_counter()
                with open(os.devnull, "wb") as dn:
                    r = subprocess.run([exe, "-c", "-l", str(args.level), "-@", str(ncores), p], stdout=dn)
                best = min(best or 1e9, time.perf_counter() - t)
            os.unlink(p)
            out["cpu_baseline"] = {"value": round(lim / best / 1e9, 3), "unit": "GB/s", "cores": ncores, "kind": "reference",
                                   "sample": f"oracle/_ref/ref_bgzip_ld -l{args.level} -@{ncores} (htslib bgzf.c + libdeflate 1.8) "
                                             f"on {lim / 2**30:.2f} GiB of the same BAM from /dev/shm, best of 2"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(2)


if __name__ == "__main__":
    main()
