"""N>1 path on CPU (gloo, world_size 2): the static block split and the cross-rank reduction that
bench.py uses.  The data path itself has no collective (blocks are independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from htslib_amd import synth


def test_static_split_is_contiguous_complete_and_balanced(built):
    from htslib_amd import _native as nat
    from htslib_amd.bgzf import shard_blocks
    _, bg = synth.bam_bgzf(6 << 20)
    desc, total = nat.bgzf_scan(bg)
    for world in (1, 2, 3, 4, 8):
        parts = shard_blocks(desc, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == len(desc)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        sizes = [int(desc["ulen"][a:b].sum()) for a, b in parts]
        assert sum(sizes) == total
        assert max(sizes) - min(sizes) <= 2 * 0xFF00 + total // 1000          # within a couple of blocks
    assert shard_blocks(desc[:0], 4) == [(0, 0)] * 4
    assert shard_blocks(desc[:1], 2)[0][1] + shard_blocks(desc[:1], 2)[1][1] >= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from htslib_amd.bgzf import reduce_timing
    import bench
    # every rank prepares its OWN shard deterministically (weak scaling: different seeds)
    seed = 0x5EED0001 + 1000003 * rank
    comp = bench.prepare(seed, 1 << 20, 6, 1, None)
    again = bench.prepare(seed, 1 << 20, 6, 1, None)
    elapsed = 0.010 * (rank + 1)
    out = reduce_timing(elapsed, float(1 << 20), float(len(comp)), comp == again, world)
    dist.barrier()
    q.put((rank, out, len(comp)))
    dist.destroy_process_group()


def test_two_rank_reduction_over_gloo(built):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, o0, c0), (r1, o1, c1) = res
    assert o0 == o1                                            # every rank sees the same reduced values
    elapsed, sum_u, sum_c, ok = o0
    assert elapsed == pytest.approx(0.020) and sum_u == 2 * (1 << 20) and sum_c == c0 + c1 and ok
    assert c0 != c1                                            # different shards, not replicas of one


def _shared_prep_worker(rank, world, cache, q):
    import bench
    bench.CHUNK = 1 << 20
    comp, start = bench.prepare_shared(rank, world, 0x5EED0001, 5 << 20, 6, 2, cache, rotate=True, timeout_s=300)
    q.put((rank, start, comp))


def test_bench_ranks_share_one_cooperatively_built_data_set(tmp_path):
    """bench.py at N > 1: the ranks build ONE synthetic BGZF data set together (rank r deflates chunks r, r+N, ...) and
    each takes all of it starting at its own chunk, instead of every rank preparing a private 10 GiB."""
    import multiprocessing as mp
    import zlib
    import bench
    from htslib_amd import synth
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    cache = str(tmp_path / "cache")
    ps = [ctx.Process(target=_shared_prep_worker, args=(r, 2, cache, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = {}
    for _ in ps:
        r, start, comp = q.get(timeout=600)
        got[r] = (start, comp)
    for p in ps:
        p.join(timeout=60)
    assert got[0][0] == 0 and got[1][0] == 2                          # 5 chunks: rank 1 starts at chunk 2
    files = sorted(os.listdir(cache))
    assert len(files) == 5 and not any(f.endswith(".tmp") for f in files)
    chunks = [open(os.path.join(cache, "v1_5eed0001_%d_%d_6.bgzf" % (i, 1 << 20)), "rb").read() for i in range(5)]
    assert got[0][1] == b"".join(chunks) and got[1][1] == b"".join(chunks[2:] + chunks[:2])
    # what rank 1 verifies against in bench.py: its first chunk regenerated from (seed, chunk index)
    plain2, _, _ = synth.bam_stream(1 << 20, 0x5EED0001, 2, False)
    out, p, c = b"", 0, chunks[2]
    while p < len(c):
        bs = int.from_bytes(c[p + 16:p + 18], "little") + 1
        out += zlib.decompress(c[p + 18:p + bs - 8], -15)
        p += bs
    assert out == plain2


def _strong_worker(rank, world, port, path, q):
    """What bench.py --scaling strong does per rank, with zlib standing in for the kernel (this test is about the SPLIT:
    one file, block ranges from shard_blocks, no exchange of data between ranks)."""
    import hashlib
    import zlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from htslib_amd import _native as nat
    from htslib_amd.bgzf import shard_blocks, reduce_timing
    comp = open(path, "rb").read()
    desc, total = nat.bgzf_scan(comp)
    lo, hi = shard_blocks(desc, world)[rank]
    out = bytearray()
    for d in desc[lo:hi]:
        a = int(d["coff"]); blk = comp[a:a + int(d["clen"])]
        piece = zlib.decompress(blk[18:-8], -15)
        assert zlib.crc32(piece) == int.from_bytes(blk[-8:-4], "little") and len(piece) == int(d["ulen"])
        out += piece
    elapsed, sum_u, sum_c, ok = reduce_timing(0.01, float(len(out)), 0.0, True, world)
    # the sink is host memory: rank outputs are simply concatenated in rank order (gather only for the check)
    parts = [None] * world
    dist.all_gather_object(parts, bytes(out))
    dist.barrier()
    q.put((rank, hashlib.md5(b"".join(parts)).hexdigest(), int(sum_u), int(total), (lo, hi)))
    dist.destroy_process_group()


def test_strong_scaling_split_reassembles_the_file(built, tmp_path):
    """bench.py --scaling strong: ONE BGZF file, rank r decodes the contiguous block range shard_blocks gives it; the
    concatenation of the rank outputs is the file's plain stream (md5 equal to the single-rank decode)."""
    import hashlib
    plain, bg = synth.bam_bgzf(3 << 20)
    path = str(tmp_path / "one.bam")
    open(path, "wb").write(bg)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = hashlib.md5(plain).hexdigest()
    for rank, md5, sum_u, total, rng in res:
        assert md5 == want and sum_u == total == len(plain)
    assert res[0][4][1] == res[1][4][0] and res[0][4][0] == 0          # contiguous, no overlap
