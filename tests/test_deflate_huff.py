"""CPU tests of the serial part of the GPU deflate encoder (htslib_amd/csrc/deflate_huff.h, compiled
for the host): length-limited Huffman code lengths, canonical codes, RFC 1951 dynamic header."""
import ctypes as C
import heapq
import os
import subprocess
import zlib

import numpy as np
import pytest

from tests import refutil

ROOT = refutil.ROOT


@pytest.fixture(scope="module")
def hh():
    so = os.path.join(ROOT, "tests", "native", "libhuffhost.so")
    src = os.path.join(ROOT, "tests", "native", "huff_host.cpp")
    hdr = os.path.join(ROOT, "htslib_amd", "csrc", "deflate_huff.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "htslib_amd", "csrc"), src, "-o", so], check=True)
    L = C.CDLL(so)
    L.hh_encode_tokens.restype = C.c_long
    L.hh_encode_tokens.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    L.hh_build_lengths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.hh_len_symbol.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hh_dist_symbol.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def optimal_cost(freq):
    h = [(f, i) for i, f in enumerate(freq) if f]
    if len(h) < 2:
        return sum(freq)
    heapq.heapify(h)
    cost = 0
    while len(h) > 1:
        a = heapq.heappop(h); b = heapq.heappop(h)
        cost += a[0] + b[0]
        heapq.heappush(h, (a[0] + b[0], -1))
    return cost


def lengths(hh, freq, maxbits):
    f = np.asarray(freq, dtype=np.uint32)
    out = np.zeros(len(f), dtype=np.uint8)
    hh.hh_build_lengths(f.ctypes.data, len(f), maxbits, out.ctypes.data)
    return out


@pytest.mark.parametrize("seed", range(12))
def test_lengths_are_complete_limited_and_optimal_when_unconstrained(hh, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([19, 30, 286]))
    kind = seed % 4
    if kind == 0: freq = rng.integers(0, 1000, n)
    elif kind == 1: freq = (rng.pareto(0.7, n) * 10).astype(np.int64)           # heavy tail -> deep trees
    elif kind == 2: freq = np.where(rng.random(n) < 0.1, rng.integers(1, 50, n), 0)
    else: freq = np.array([int(1.6 ** i) for i in range(n)]) % (1 << 31)         # fibonacci-like: needs limiting
    freq = np.asarray(freq, dtype=np.int64)
    maxbits = 7 if n == 19 else 15
    ln = lengths(hh, freq, maxbits)
    used = freq > 0
    assert ln.max() <= maxbits
    if used.sum() >= 2:
        assert (ln[used] > 0).all() and (ln[~used] == 0).all()
        assert sum(2.0 ** -int(l) for l in ln if l) == 1.0                       # complete code
        cost = int((freq * ln).sum())
        opt = optimal_cost(freq.tolist())
        assert cost >= opt
        unl = lengths(hh, freq, 30)
        if unl.max() <= maxbits:
            assert cost == opt
        else:
            assert cost <= opt * 1.05 + 64
    else:
        assert sum(2.0 ** -int(l) for l in ln if l) == 1.0                       # two 1-bit codes


def test_symbol_mapping_matches_rfc1951_tables(hh):
    lbase = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
    lext = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
    dbase = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
    dext = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
    s, xb, xv = C.c_uint32(), C.c_uint32(), C.c_uint32()
    for ln in range(3, 259):
        hh.hh_len_symbol(ln, C.byref(s), C.byref(xb), C.byref(xv))
        k = max(i for i in range(29) if lbase[i] <= ln) if ln != 258 else 28
        assert (s.value, xb.value, xv.value) == (k, lext[k], ln - lbase[k])
    for d in list(range(1, 2000)) + [4096, 4097, 8192, 16384, 24576, 24577, 32767, 32768]:
        hh.hh_dist_symbol(d, C.byref(s), C.byref(xb), C.byref(xv))
        k = max(i for i in range(30) if dbase[i] <= d)
        assert (s.value, xb.value, xv.value) == (k, dext[k], d - dbase[k])


@pytest.mark.parametrize("case", ["text", "binary", "one_symbol", "empty", "runs"])
def test_dynamic_header_and_codes_decode_with_zlib(hh, oracle, case):
    rng = np.random.default_rng(5)
    data = {"text": b"GATTACA quality IIIIFFFF:::: " * 400, "binary": rng.integers(0, 256, 20000, dtype=np.uint8).tobytes(),
            "one_symbol": b"A" * 5000, "empty": b"", "runs": bytes([7] * 3000 + [9] * 10 + list(range(256)) * 3)}[case]
    tok = np.frombuffer(data, dtype=np.uint8).astype(np.uint32)
    if case in ("text", "runs") and len(data) > 600:            # sprinkle real matches over repeated text
        toks, i = [], 0
        while i < len(data):
            per = 29 if case == "text" else 1
            if i >= 300 and i + 40 < len(data) and data[i:i + 40] == data[i - per:i - per + 40] and rng.random() < 0.5:
                ln = int(rng.integers(3, 41)); toks.append(0x80000000 | ((ln - 3) << 16) | (per - 1)); i += ln
            else:
                toks.append(data[i]); i += 1
        tok = np.array(toks, dtype=np.uint32)
    out = np.zeros(len(data) * 2 + 1024, dtype=np.uint8)
    n = hh.hh_encode_tokens(tok.ctypes.data, len(tok), out.ctypes.data, len(out))
    assert n > 0
    raw = out[:n].tobytes()
    assert zlib.decompress(raw, -15) == data
    rc, got, used = oracle.inflate_raw(raw, len(data) + 16)
    assert rc == 0 and got == data


# ---- the workgroup-collective version (htslib_amd/csrc/deflate_huff_wg.h): package-merge lengths, parallel header ----------------
@pytest.fixture(scope="module")
def hhw():
    so = os.path.join(ROOT, "tests", "native", "libhuffwghost.so")
    src = os.path.join(ROOT, "tests", "native", "huffwg_host.cpp")
    hdrs = [os.path.join(ROOT, "htslib_amd", "csrc", h) for h in ("deflate_huff.h", "deflate_huff_wg.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "htslib_amd", "csrc"), src, "-o", so, "-lpthread"], check=True)
    L = C.CDLL(so)
    L.hhw_encode_tokens.restype = C.c_long
    L.hhw_encode_tokens.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int]
    L.hhw_build_lengths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return L


def limited_optimal_cost(freq, maxbits):
    """Exact optimum of sum(freq * len) under len <= maxbits (Kraft equality): textbook package-merge; an item carries the summed
    weight of the leaves inside it counted with multiplicity, i.e. what it adds to the cost when selected."""
    w = sorted(f for f in freq if f)
    n = len(w)
    if n < 2:
        return sum(freq)
    prev = [(x, x) for x in w]
    for _ in range(maxbits - 1):
        pk = [(prev[2 * j][0] + prev[2 * j + 1][0], prev[2 * j][1] + prev[2 * j + 1][1]) for j in range(len(prev) // 2)]
        prev = sorted([(x, x) for x in w] + pk, key=lambda t: t[0])
    return sum(t[1] for t in prev[:2 * n - 2])


def wg_lengths(hhw, freq, maxbits, nt=32):
    f = np.asarray(freq, dtype=np.uint32)
    out = np.zeros(len(f), dtype=np.uint8)
    hhw.hhw_build_lengths(f.ctypes.data, len(f), maxbits, out.ctypes.data, nt)
    return out


def test_package_merge_list_sizes_reach_the_selection_point():
    # the walk starts at item 2n - 2 of the last level's list: the list must be that long (15 levels for <= 286 leaves, 7 for <= 19)
    for levels, nmax in ((15, 286), (7, 19)):
        for n in range(2, nmax + 1):
            m = n
            for _ in range(levels - 1):
                m = n + m // 2
            assert m >= 2 * n - 2, (levels, n, m)


@pytest.mark.parametrize("seed", range(24))
def test_wg_lengths_are_complete_limited_and_exactly_optimal(hhw, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([19, 30, 286]))
    kind = seed % 6
    if kind == 0: freq = rng.integers(0, 1000, n)
    elif kind == 1: freq = (rng.pareto(0.7, n) * 10).astype(np.int64)
    elif kind == 2: freq = np.where(rng.random(n) < 0.1, rng.integers(1, 50, n), 0)
    elif kind == 3: freq = np.array([int(1.6 ** i) for i in range(n)]) % (1 << 30)
    elif kind == 4: freq = np.ones(n, dtype=np.int64) * int(rng.integers(1, 5))     # all ties
    else: freq = np.where(np.arange(n) < int(rng.integers(0, 3)), 7, 0)              # 0, 1 or 2 used symbols
    freq = np.minimum(np.asarray(freq, dtype=np.int64), 1 << 24)
    maxbits = 7 if n == 19 else 15
    ln = wg_lengths(hhw, freq, maxbits, nt=256 if seed % 8 == 0 else 32)
    used = freq > 0
    assert ln.max() <= maxbits
    assert sum(2.0 ** -int(l) for l in ln if l) == 1.0
    if used.sum() >= 2:
        assert (ln[used] > 0).all() and (ln[~used] == 0).all()
        assert int((freq * ln).sum()) == limited_optimal_cost(freq.tolist(), maxbits)


@pytest.mark.parametrize("nt", [32, 256])
@pytest.mark.parametrize("case", ["text", "binary", "one_symbol", "empty", "runs", "long_zero_runs", "repeats"])
def test_wg_dynamic_header_and_codes_decode_with_zlib(hh, hhw, oracle, case, nt):
    rng = np.random.default_rng(5)
    data = {"text": b"GATTACA quality IIIIFFFF:::: " * 400, "binary": rng.integers(0, 256, 20000, dtype=np.uint8).tobytes(),
            "one_symbol": b"A" * 5000, "empty": b"", "runs": bytes([7] * 3000 + [9] * 10 + list(range(256)) * 3),
            "long_zero_runs": bytes([0, 255] * 2000 + [3] * 50),                 # symbols 1..254 unused: zero runs > 138 in the header
            "repeats": bytes(rng.permutation(256).astype(np.uint8).tolist() * 40)}[case]   # equal frequencies: long runs of one length
    tok = np.frombuffer(data, dtype=np.uint8).astype(np.uint32)
    if case in ("text", "runs") and len(data) > 600:
        toks, i = [], 0
        while i < len(data):
            per = 29 if case == "text" else 1
            if i >= 300 and i + 40 < len(data) and data[i:i + 40] == data[i - per:i - per + 40] and rng.random() < 0.5:
                ln = int(rng.integers(3, 41)); toks.append(0x80000000 | ((ln - 3) << 16) | (per - 1)); i += ln
            else:
                toks.append(data[i]); i += 1
        tok = np.array(toks, dtype=np.uint32)
    out = np.zeros(len(data) * 2 + 1024, dtype=np.uint8)
    n = hhw.hhw_encode_tokens(tok.ctypes.data, len(tok), out.ctypes.data, len(out), nt)
    assert n > 0
    raw = out[:n].tobytes()
    assert zlib.decompress(raw, -15) == data
    rc, got, used = oracle.inflate_raw(raw, len(data) + 16)
    assert rc == 0 and got == data
    # never larger than the serial construction (package-merge is optimal, the token coding of the header is the same greedy scan)
    out2 = np.zeros(len(out), dtype=np.uint8)
    n2 = hh.hh_encode_tokens(tok.ctypes.data, len(tok), out2.ctypes.data, len(out2))
    assert n <= n2


def test_input_ring_schedule_of_the_deflate_kernel_never_overwrites_live_bytes():
    """bgzf_deflate.hip stages the block as a ring of RING bytes: before chunk c0 the bytes up to c0 + AHEAD must be there, and the REFILL bytes a chunk requests
    for the next one may only replace positions the next chunk can no longer name: a candidate lies at most WINDOW + 1 before its position (round 6: the window is
    what the ring leaves, 19 680 bytes, not DEFLATE's 32 KiB -- the hash table does not remember farther back).  The constants are read from the kernel source; the
    schedule is the kernel's (`refill = hi < pad_end && hi < c0 + WG + AHEAD`)."""
    import re
    src = open(os.path.join(ROOT, "htslib_amd", "csrc", "bgzf_deflate.hip")).read()
    m = re.search(r"constexpr uint32_t RING = (\d+)u, MIRROR = (\d+)u, REFILL = (\d+)u, AHEAD = (\d+)u;", src)
    assert m, "ring constants not found"
    RING, MIRROR, REFILL, AHEAD = map(int, m.groups())
    WG = 256
    assert "const bool refill = hi < pad_end && hi < c0 + WG + AHEAD;" in src
    assert "constexpr uint32_t WINDOW = RING - (WG + AHEAD + REFILL);" in src and "const uint32_t dcap = p < WINDOW + 1u ? p : WINDOW + 1u;" in src
    WINDOW = RING - (WG + AHEAD + REFILL)
    assert RING % 32 == 0 and REFILL == WG * 8 and MIRROR >= 36 + 4 and 16384 <= WINDOW <= 32768 and 3 * RING >= 65280 + 64   # a compare reads 36 bytes from a dword-aligned address
    for n in list(range(1, 600, 37)) + list(range(RING - 80, RING + 80, 7)) + list(range(40000, 65281, 211)) + [65280]:
        pad_end = (n + 48 + 15) & ~15
        hi = min(pad_end, RING)
        for c0 in range(0, n, WG):
            last_read = min(n + 19, c0 + 255 + 258 + 19)                     # own string, fixed compare (p + 35), long-match rounds (p + l + 19, l < 258)
            assert hi > last_read or hi >= pad_end, (n, c0, hi)
            if hi < pad_end and hi < c0 + WG + AHEAD:
                replaced_end = hi + REFILL - RING                            # positions [hi - RING, replaced_end) lose their bytes
                assert replaced_end <= max(0, c0 + WG - (WINDOW + 1)), (n, c0, hi)  # dead for the next chunk (and for this chunk's literal reads: < c0)
                hi += REFILL
