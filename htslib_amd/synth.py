"""Seeded synthetic workloads for the block-codec hot path.

WORKLOAD PREPARATION ONLY.  Nothing here is on the product path: it builds the
*inputs* of the benchmarks/tests (a coordinate-sorted 150 bp paired-end BAM
byte stream, cut into BGZF blocks the way ``bam_write1`` + ``bgzf_flush_try``
do, reference sam.c:862-900 / bgzf.c:1996-2000), and, for the inflate
workloads, deflates them on the host with the *system zlib* through Python's
``zlib`` module using exactly the parameters of the reference's zlib path
(``deflateInit2(level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY)``,
bgzf.c:647) so the streams the GPU inflates are the streams stock htslib
writes.

All randomness comes from ``numpy.random.Generator(PCG64(seed))`` with
seed = 0x5EED0001 + chunk index, so every chunk is reproducible on its own.
"""
from __future__ import annotations

import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

SEED = 0x5EED0001
BGZF_BLOCK_SIZE = 0xFF00
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_BGZF_HDR = bytes.fromhex("1f8b08040000000000ff060042430200")

N_REF = 25
REF_LEN = 100_000_000
READ_LEN = 150


# --------------------------------------------------------------------------- BAM
def bam_header() -> bytes:
    """BAM header with 25 @SQ lines (binary form, SAM spec 4.2)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n"
    for i in range(N_REF):
        text += f"@SQ\tSN:chr{i + 1}\tLN:{REF_LEN}\n"
    text += "@RG\tID:grp1\tSM:synthetic\tPL:ILLUMINA\n"
    tb = text.encode()
    out = [b"BAM\x01", struct.pack("<i", len(tb)), tb, struct.pack("<i", N_REF)]
    for i in range(N_REF):
        nm = f"chr{i + 1}".encode() + b"\0"
        out += [struct.pack("<i", len(nm)), nm, struct.pack("<i", REF_LEN)]
    return b"".join(out)


def _digits(v: np.ndarray, width: int):
    """Left-aligned decimal digits of v in a [N,width] uint8 matrix + digit count."""
    v = v.astype(np.int64)
    nd = np.ones(v.shape, dtype=np.int64)
    for k in range(1, width):
        nd += v >= 10 ** k
    out = np.zeros((v.shape[0], width), dtype=np.uint8)
    for c in range(width):
        p = nd - 1 - c
        d = (v // np.power(10, np.maximum(p, 0))) % 10
        out[:, c] = np.where(p >= 0, d + 48, 0)
    return out, nd


def _reg2bin(beg: np.ndarray, end: np.ndarray) -> np.ndarray:
    """UCSC binning scheme (SAM spec 5.3)."""
    end = end - 1
    b = np.zeros_like(beg)
    done = np.zeros(beg.shape, dtype=bool)
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        m = ~done & ((beg >> shift) == (end >> shift))
        b = np.where(m, off + (beg >> shift), b)
        done |= m
    return b


QUAL_VARIANT = "A"       # "A": NovaSeq-like 4-bin qualities (SURVEY 8d default); "B": HiSeq-like 41-level qualities (mean 36 decaying to 28 along the
                         # read, sigma 4, clamp 2..41) -- compresses ~3.3x instead of ~5.5x.  Set before the worker pool forks.


def bam_records(n: int, seed: int = SEED, chunk: int = 0) -> bytes:
    """n coordinate-sorted paired 150 bp alignment records as BAM bytes.

    Reads are drawn from a per-chunk random genome at ~30x coverage (so that
    neighbouring records overlap, as in a real sorted BAM), 0.5 % substitution
    errors, 4-bin NovaSeq-like qualities {2,12,23,37} as a Markov chain
    (stay 0.9, start 37), CIGAR 150M (95 %) or aSbM (5 %), MAPQ {0,60},
    tags NM:C, MD:Z, RG:Z.
    """
    rng = np.random.Generator(np.random.PCG64(seed + chunk))
    L = READ_LEN
    # ---- coordinates -----------------------------------------------------
    step = rng.geometric(0.2, size=n).astype(np.int64) - 1            # mean 4 bp -> ~37x
    gpos = np.cumsum(step)
    span = int(gpos[-1]) + L + 1
    per_ref = span // N_REF + 1
    refid = (gpos // per_ref).astype(np.int32)
    pos = (gpos - refid.astype(np.int64) * per_ref + 10_000 + chunk * 1000).astype(np.int32)
    genome = rng.integers(0, 4, size=span + L, dtype=np.uint8)
    # ---- sequence ----------------------------------------------------------
    idx = gpos[:, None] + np.arange(L, dtype=np.int64)[None, :]
    bases = genome[idx]                                                   # 0..3
    err = rng.random((n, L)) < 0.005
    nerr = err.sum(axis=1)
    bases = np.where(err, (bases + rng.integers(1, 4, size=(n, L), dtype=np.uint8)) & 3, bases)
    isn = rng.random((n, L)) < 0.002
    code = np.where(isn, 15, np.array([1, 2, 4, 8], dtype=np.uint8)[bases]).astype(np.uint8)
    seq = (code[:, 0::2] << 4) | code[:, 1::2]                            # [n, 75]
    # ---- qualities: 4-state Markov chain ---------------------------------
    change = rng.random((n, L)) < 0.1
    change[:, 0] = True
    newq = np.array([2, 12, 23, 37], dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    newq[:, 0] = 37
    ci = np.where(change, np.arange(L)[None, :], 0)
    ci = np.maximum.accumulate(ci, axis=1)
    qual = np.take_along_axis(newq, ci, axis=1)
    if QUAL_VARIANT == "B":
        mean = 36.0 - 8.0 * np.arange(L, dtype=np.float32)[None, :] / (L - 1)
        qual = np.clip(np.rint(mean + 4.0 * rng.standard_normal((n, L), dtype=np.float32)), 2, 41).astype(np.uint8)
    # ---- fixed fields --------------------------------------------------------
    mapq = np.where(rng.random(n) < 0.05, 0, 60).astype(np.uint8)
    flag = np.array([99, 147, 83, 163], dtype=np.uint16)[rng.integers(0, 4, size=n)]
    isz = rng.integers(250, 500, size=n).astype(np.int32)
    fwd = (flag == 99) | (flag == 163)
    mpos = np.where(fwd, pos + isz - L, np.maximum(pos - isz + L, 0)).astype(np.int32)
    tlen = np.where(fwd, isz, -isz).astype(np.int32)
    clip = np.where(rng.random(n) < 0.05, rng.integers(1, 51, size=n), 0).astype(np.uint32)
    ncig = 1 + (clip > 0)
    bin_ = _reg2bin(pos.astype(np.int64), pos.astype(np.int64) + L - clip).astype(np.uint16)
    # ---- read names: SIM:1:FC01:<lane>:<tile>:<x>:<y> ----------------------
    lane = rng.integers(1, 9, size=n)
    tile = rng.integers(1101, 2679, size=n)
    xs = rng.integers(1000, 32768, size=n)
    ys = rng.integers(1000, 99999, size=n)
    # ---- row matrix + validity mask ---------------------------------------
    W = 36 + 40 + 8 + 75 + L + 4 + 16 + 8
    mat = np.zeros((n, W), dtype=np.uint8)
    msk = np.zeros((n, W), dtype=bool)
    col = 0

    def put(block: np.ndarray, valid=None):
        nonlocal col
        w = block.shape[1]
        mat[:, col:col + w] = block
        msk[:, col:col + w] = True if valid is None else valid
        col += w

    def le(v: np.ndarray, nbytes: int) -> np.ndarray:
        v = v.astype(np.int64) & ((1 << (8 * nbytes)) - 1)
        return np.stack([(v >> (8 * k)) & 0xFF for k in range(nbytes)], axis=1).astype(np.uint8)

    fixed_col = col
    put(np.zeros((n, 36), dtype=np.uint8))                                # patched below
    prefix = np.frombuffer(b"SIM:1:FC01:", dtype=np.uint8)
    put(np.broadcast_to(prefix, (n, len(prefix))))
    put((lane + 48).astype(np.uint8)[:, None])
    put(np.full((n, 1), ord(":"), dtype=np.uint8))
    d, _ = _digits(tile, 4); put(d)
    put(np.full((n, 1), ord(":"), dtype=np.uint8))
    d, ndx = _digits(xs, 5); put(d, d != 0)
    put(np.full((n, 1), ord(":"), dtype=np.uint8))
    d, ndy = _digits(ys, 5); put(d, d != 0)
    put(np.zeros((n, 1), dtype=np.uint8))                                 # NUL
    l_read_name = (len(prefix) + 1 + 1 + 4 + 1 + ndx + 1 + ndy + 1).astype(np.uint8)
    # cigar
    soft = le((clip << 4) | 4, 4)
    put(soft, np.broadcast_to((clip > 0)[:, None], (n, 4)))
    put(le(((L - clip) << 4) | 0, 4))
    put(seq)
    put(qual)
    # aux: NM:C
    put(np.broadcast_to(np.frombuffer(b"NMC", dtype=np.uint8), (n, 3)))
    put(np.minimum(nerr, 255).astype(np.uint8)[:, None])
    # aux: MD:Z  ("150" or "<m><base><149-m>" for the first mismatch)
    put(np.broadcast_to(np.frombuffer(b"MDZ", dtype=np.uint8), (n, 3)))
    has = nerr > 0
    first = np.where(has, err.argmax(axis=1), L)
    d1, n1 = _digits(first, 3)
    put(d1, np.arange(3)[None, :] < n1[:, None])
    refbase = np.array([65, 67, 71, 84], dtype=np.uint8)[genome[gpos + np.minimum(first, L - 1)]]
    put(refbase[:, None], has[:, None])
    rest = np.where(has, L - 1 - first, 0)
    d2, n2 = _digits(rest, 3)
    put(d2, has[:, None] & (np.arange(3)[None, :] < n2[:, None]))
    put(np.zeros((n, 1), dtype=np.uint8))
    # aux: RG:Z:grp1
    put(np.broadcast_to(np.frombuffer(b"RGZgrp1\0", dtype=np.uint8), (n, 8)))
    assert col <= W
    # ---- fixed 36-byte core (SAM spec 4.2) ---------------------------------
    block_size = msk.sum(axis=1).astype(np.int64) - 4
    core = np.concatenate([
        le(block_size, 4), le(refid, 4), le(pos, 4), l_read_name[:, None], mapq[:, None],
        le(bin_, 2), le(ncig, 2), le(flag, 2), le(np.full(n, L), 4), le(refid, 4), le(mpos, 4), le(tlen, 4)
    ], axis=1)
    mat[:, fixed_col:fixed_col + 36] = core
    return mat[msk].tobytes()


def bam_stream(nbytes: int, seed: int = SEED, chunk: int = 0, with_header: bool = True):
    """(data, record_starts): >= nbytes of BAM (header + whole records).

    record_starts are byte offsets at which a BGZF block may be cut (the
    header counts as one unit, like bam_hdr_write + bgzf_flush)."""
    hdr = bam_header() if with_header else b""
    n = max(16, int(nbytes / 308.6) + 8)          # mean record = 308.6 B
    body = bam_records(n, seed, chunk)
    a = np.frombuffer(body, dtype=np.uint8)
    # record boundaries by walking block_size (vectorised: sizes differ little, walk in python once)
    starts = []
    p = 0
    ln = len(body)
    mv = memoryview(body)
    unpack = struct.unpack_from
    while p < ln:
        starts.append(p)
        p += 4 + unpack("<i", mv, p)[0]
    assert p == ln
    starts = np.asarray(starts, dtype=np.int64) + len(hdr)
    del a
    return hdr + body, starts, len(hdr)


def cut_blocks(total_len: int, rec_starts: np.ndarray, hdr_len: int):
    """Block boundaries exactly as bam_write1 + bgzf_flush_try produce them:
    records are kept whole, a block is flushed when the next record would take
    it past BGZF_BLOCK_SIZE (bgzf.c:1996-2000); the header is flushed alone."""
    cuts = [0]
    if hdr_len:
        # header written through bgzf_write: split every 0xff00 bytes, then flushed
        p = 0
        while hdr_len - p > BGZF_BLOCK_SIZE:
            p += BGZF_BLOCK_SIZE
            cuts.append(p)
        cuts.append(hdr_len)
    ends = np.append(rec_starts[1:], total_len)
    cur = cuts[-1]
    i = 0
    nrec = len(rec_starts)
    while i < nrec:
        # largest j with ends[j-1] - cur <= BLOCK_SIZE
        j = int(np.searchsorted(ends, cur + BGZF_BLOCK_SIZE, side="right"))
        if j <= i:          # a single record larger than a block: split inside bgzf_write
            cur += BGZF_BLOCK_SIZE
            cuts.append(cur)
            continue
        cur = int(ends[j - 1])
        cuts.append(cur)
        i = j
    return np.asarray(cuts, dtype=np.int64)


# -------------------------------------------------------------------------- FASTQ
def fastq(nbytes: int, seed: int = SEED, chunk: int = 0) -> bytes:
    """Synthetic FASTQ (config C1): @SIM:1:FC01:<lane>:<tile>:<x>:<y> 1:N:0:ACGTACGT."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919 + chunk))
    n = max(4, nbytes // 350 + 1)
    L = READ_LEN
    u = rng.random((n, L))
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[(u * 4).astype(np.int64) & 3].copy()
    seq[rng.random((n, L)) < 0.002] = ord("N")
    change = rng.random((n, L)) < 0.1
    change[:, 0] = True
    newq = np.array([2, 12, 23, 37], dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    newq[:, 0] = 37
    ci = np.maximum.accumulate(np.where(change, np.arange(L)[None, :], 0), axis=1)
    qual = np.take_along_axis(newq, ci, axis=1) + 33
    lane = rng.integers(1, 9, size=n); tile = rng.integers(1101, 2679, size=n)
    xs = rng.integers(1000, 32768, size=n); ys = rng.integers(1000, 99999, size=n)
    out = []
    for i in range(n):
        out.append(b"@SIM:1:FC01:%d:%d:%d:%d 1:N:0:ACGTACGT\n" % (lane[i], tile[i], xs[i], ys[i]))
        out.append(seq[i].tobytes()); out.append(b"\n+\n"); out.append(qual[i].tobytes()); out.append(b"\n")
    return b"".join(out)


# ------------------------------------------------------------ host-side BGZF writer
def bgzf_block(data: bytes, level: int = 6) -> bytes:
    """One BGZF block the way the reference's zlib path writes it (bgzf.c:624-683)."""
    if len(data) == 0:
        return BGZF_EOF
    if level == 0:
        payload = b"\x01" + struct.pack("<HH", len(data), len(data) ^ 0xFFFF) + data
    else:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY)
        payload = co.compress(data) + co.flush()
        if len(payload) + 26 > 0x10000:       # would not fit: stored block (bgzf.c:652-667)
            payload = b"\x01" + struct.pack("<HH", len(data), len(data) ^ 0xFFFF) + data
    bsize = len(payload) + 26
    return b"".join([_BGZF_HDR, struct.pack("<H", bsize - 1), payload,
                     struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))])


def bgzf_compress(data: bytes, cuts=None, level: int = 6, threads: int = 1, eof: bool = True) -> bytes:
    """Deflate `data` into a BGZF stream; blocks end at `cuts` (default: every 0xff00 bytes)."""
    if cuts is None:
        cuts = list(range(0, len(data), BGZF_BLOCK_SIZE)) + [len(data)]
        if len(data) == 0:
            cuts = [0]
    mv = memoryview(data)
    spans = [(int(cuts[i]), int(cuts[i + 1])) for i in range(len(cuts) - 1)]

    def work(se):
        return bgzf_block(bytes(mv[se[0]:se[1]]), level)

    if threads > 1 and len(spans) > 8:
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(work, spans, chunksize=16))
    else:
        parts = [work(s) for s in spans]
    if eof:
        parts.append(BGZF_EOF)
    return b"".join(parts)


def bam_bgzf(nbytes: int, seed: int = SEED, chunk: int = 0, level: int = 6, threads: int = 1,
             with_header: bool = True, eof: bool = True):
    """(plain_bam_bytes, bgzf_bytes) for ~nbytes of synthetic BAM."""
    data, starts, hdr_len = bam_stream(nbytes, seed, chunk, with_header)
    cuts = cut_blocks(len(data), starts, hdr_len)
    return data, bgzf_compress(data, cuts, level, threads, eof)
