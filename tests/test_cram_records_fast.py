"""The data-parallel CRAM record decoder (htslib_amd/csrc/cram_records_fast.h; reference cram/cram_decode.c:2553-2985, :2140-2307): per-record
passes + prefix sums instead of one serial chain per slice.  The per-record passes are ONE source for host and device; here the CPU compile
(tests/native/cram_records_host.cpp, plain loops in the kernels' order) is checked against

  * the SAM / BAM twins of the reference's 34 CRAM fixtures (slices the path takes; the rest must fall through to the chain decoder),
  * the pinned chain decoder, column for column, on production-shaped synthetic slices (tags, MD / NM regeneration, unmapped and
    detached records, multi-slice batches),
  * the chain decoder's verdict on damaged slices (the path may only ever say "not mine").

The -m gpu tests run the same comparisons through the kernels (tests/test_cram_records.py holds the shared plumbing)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import test_cram_records as T

_vp = C.c_void_p


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    import subprocess
    so = str(tmp_path_factory.mktemp("cramfast") / "libcram_records_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas"] + os.environ.get("HG_TEST_HOSTLIB_FLAGS", "").split() + ["-o", so, os.path.join(T.ROOT, "tests", "native", "cram_records_host.cpp")], check=True)
    L = C.CDLL(so)
    L.hgr_host_records_bound.argtypes = [C.c_size_t, _vp, C.c_int, _vp, _vp, _vp, _vp]
    L.hgr_host_decode_records.argtypes = [C.c_size_t, _vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _vp, _vp, _vp]
    L.hgr_host_decode_records_fast.argtypes = L.hgr_host_decode_records.argtypes + [_vp]
    return L


def fast_call(L):
    """hgr_host_decode_records_fast with decode()'s calling convention; .path = which slices the data-parallel path decoded"""
    def call(n, *a):
        path = np.full(max(n, 1), -1, np.int32)
        rc = L.hgr_host_decode_records_fast(n, *a, path.ctypes.data)
        call.path = path[:n].copy()
        return rc
    return call


def test_fixture_slices_the_path_takes_match_the_sam_twins(hostlib):
    files = {}
    for fname, major, nref, s in T.load_slices():
        files.setdefault((fname, major, nref), []).append(s)
    fc = fast_call(hostlib)
    nrec = took = 0
    for (fname, major, nref), slices in files.items():
        st, got = T.decode(hostlib.hgr_host_records_bound, fc, slices, major, nref)
        assert (st == 0).all(), (fname, st)
        took += int(fc.path.sum())
        for s, g in zip(slices, got):
            T.check_against_twin(fname, g, s["expect"]); nrec += len(g)
    assert nrec == 230
    # the fixtures are CRAM 3.0 by old htslib / htsjdk writers: most put BF, CF, ... into the CORE block.  Whatever the count, every slice
    # decoded to its twin above; the synthetic tests below are the ones that must run on the path.
    print("fixture slices decoded by the data-parallel path:", took)


def _raw(call_bound, call_decode, slices, major, nref, with_seq=True):
    st, got = T.decode(call_bound, call_decode, slices, major, nref, with_seq)
    return st, got, T.decode.last_aend


@pytest.mark.parametrize("decode_md", [-1, 0])
def test_synthetic_slices_equal_the_chain_decoder(hostlib, decode_md):
    from htslib_amd import synth_cram
    rng = np.random.default_rng(101)
    slices = [synth_cram.make_slice(rng, 3000, 100), synth_cram.make_slice(rng, 500, 151, unmapped_every=3, detached_every=4), synth_cram.make_slice(rng, 1, 40, ref_len=500),
              synth_cram.make_slice(rng, 257, 75, unmapped_every=0, detached_every=0), synth_cram.make_slice(rng, 1200, 100, tags=True),
              synth_cram.make_slice(rng, 33, 60, unmapped_every=2, tags=True), synth_cram.make_slice(rng, 2, 50, ref_len=900), synth_cram.make_slice(rng, 255, 90),
              synth_cram.make_slice(rng, 256, 90), synth_cram.make_slice(rng, 513, 64, tags=True)]
    fc = fast_call(hostlib)
    T.DECODE_MD[0] = decode_md
    try:
        st, chain, aend = _raw(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 1)
        st2, fast, aend2 = _raw(hostlib.hgr_host_records_bound, fc, slices, 3, 1)
        st3, cols, _ = _raw(hostlib.hgr_host_records_bound, fc, slices, 3, 1, with_seq=False)
    finally:
        T.DECODE_MD[0] = -1
    assert (st == 0).all() and (st2 == 0).all() and (st3 == 0).all()
    assert fc.path.all(), fc.path                                       # every one of them went through the data-parallel passes
    assert fast == chain and aend == aend2
    assert [[r[:9] for r in s] for s in cols] == [[r[:9] for r in s] for s in chain]
    T._check_truth(slices, fast) if decode_md == 0 else None


def test_mixed_batch_fixture_and_synthetic_slices(hostlib):
    """one batch holding slices of both kinds: the CORE-coded ones keep the chain decoder, the others take the passes; placement of bases is
    deterministic (fast slices first, in slice order)"""
    from htslib_amd import synth_cram
    rng = np.random.default_rng(5)
    fx = [s for f, major, nref, s in T.load_slices() if f == "test/range.cram"]
    syn = [synth_cram.make_slice(rng, 300, 80), synth_cram.make_slice(rng, 90, 80, tags=True)]
    slices = [syn[0]] + fx[:1] + [syn[1]] + fx[1:]
    fc = fast_call(hostlib)
    st, chain, _ = _raw(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 7)
    st2, fast, _ = _raw(hostlib.hgr_host_records_bound, fc, slices, 3, 7)
    assert (st == 0).all() and (st2 == 0).all()
    assert fast == chain
    assert fc.path[0] == 1 and fc.path[2] == 1


def test_damaged_slices_get_the_chain_decoders_verdict(hostlib):
    """mutated EXTERNAL-only slices: the data-parallel path either decodes exactly what the chain decoder decodes or steps aside"""
    from htslib_amd import synth_cram
    rng = np.random.default_rng(404)
    base = [synth_cram.make_slice(rng, 60, 70, tags=True), synth_cram.make_slice(rng, 45, 50, unmapped_every=4), synth_cram.make_slice(rng, 30, 64, detached_every=2, tags=True)]
    fc = fast_call(hostlib)

    def mutate(b):
        b = bytearray(b)
        if not b: return bytes(b)
        k = int(rng.integers(0, 4))
        if k == 0:
            for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1: b = b[:int(rng.integers(0, len(b)))]
        elif k == 2: b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:
            i = int(rng.integers(0, len(b))); b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8))
        return bytes(b)

    same = took = failed = 0
    for it in range(500):
        s = dict(base[int(rng.integers(0, len(base)))])
        what = int(rng.integers(0, 6))
        if what == 0: s["comp_hdr"] = mutate(s["comp_hdr"])
        elif what == 1: s["slice_hdr"] = mutate(s["slice_hdr"][:3]) + s["slice_hdr"][3:] if it % 2 else mutate(s["slice_hdr"])
        elif what <= 4:
            j = int(rng.integers(0, len(s["blocks"]))); bl = list(s["blocks"]); bl[j] = (bl[j][0], mutate(bl[j][1])); s["blocks"] = bl
        else: s["refs"] = [(t, a, b[:len(b) // 2], ln) for t, a, b, ln in s.get("refs", [])]
        res = []
        for call in (hostlib.hgr_host_decode_records, fc):
            try:
                st, got, aend = _raw(hostlib.hgr_host_records_bound, call, [s], 3, 1)
                res.append((int(st[0]), got if st[0] == 0 else None, aend if st[0] == 0 else None))
            except AssertionError:
                res.append(("refused",))
            except (ValueError, IndexError, UnicodeDecodeError, T.struct_error):      # tag bytes that are not BAM aux: "decoded", but the text helper cannot render them
                res.append(("unrenderable",))
        assert res[0] == res[1], (it, what, res[0][0], res[1][0])
        same += 1
        if res[1][0] == 0: took += int(fc.path[0])
        else: failed += 1
    assert took > 50 and failed > 50, (took, failed)


# ---------------------------------------------------------------- on the MI355X: the kernels of cram_records_fast.hip ----
def _synthetic(rng):
    from htslib_amd import synth_cram
    return [synth_cram.make_slice(rng, 3000, 100), synth_cram.make_slice(rng, 500, 151, unmapped_every=3, detached_every=4), synth_cram.make_slice(rng, 1, 40, ref_len=500),
            synth_cram.make_slice(rng, 257, 75, unmapped_every=0, detached_every=0), synth_cram.make_slice(rng, 1200, 100, tags=True),
            synth_cram.make_slice(rng, 33, 60, unmapped_every=2, tags=True), synth_cram.make_slice(rng, 2, 50, ref_len=900), synth_cram.make_slice(rng, 255, 90),
            synth_cram.make_slice(rng, 256, 90), synth_cram.make_slice(rng, 513, 64, tags=True), synth_cram.make_slice(rng, 1025, 70), synth_cram.make_slice(rng, 4097, 36, tags=True)]


@pytest.mark.gpu
@pytest.mark.parametrize("decode_md", [-1, 0])
def test_gpu_passes_equal_the_chain_decoder_on_synthetic_slices(engine, hostlib, decode_md):
    """the default device path (data-parallel passes) == the chain decoder compiled for the CPU == the chain kernel (HG_CRAM_RECORDS_PATH=chain)"""
    bound, dec = T._gpu_calls(engine)
    slices = _synthetic(np.random.default_rng(202))
    T.DECODE_MD[0] = decode_md
    try:
        st_c, chain, aend_c = _raw(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 1)
        st_g, gpu, aend_g = _raw(bound, dec, slices, 3, 1)
        st_n, cols, _ = _raw(bound, dec, slices, 3, 1, with_seq=False)
        os.environ["HG_CRAM_RECORDS_PATH"] = "chain"
        try:
            st_k, kern, aend_k = _raw(bound, dec, slices, 3, 1)
        finally:
            del os.environ["HG_CRAM_RECORDS_PATH"]
    finally:
        T.DECODE_MD[0] = -1
    assert (st_c == 0).all() and (st_g == 0).all() and (st_n == 0).all() and (st_k == 0).all()
    assert gpu == chain and aend_g == aend_c
    assert kern == chain and aend_k == aend_c
    assert [[r[:9] for r in s] for s in cols] == [[r[:9] for r in s] for s in chain]


@pytest.mark.gpu
def test_gpu_staged_batch_gives_the_same_bam_stream_and_uses_the_passes(engine):
    """hg_cram_batch_stage + hg_cram_batch_decode_bam_dev (BAM left in HBM) == hg_cram_decode_bam_host; every synthetic slice goes through the
    passes; the stream is identical from run to run (bases are placed by prefix sums) and identical to the chain kernel's"""
    from htslib_amd import _native as nat
    slices = _synthetic(np.random.default_rng(303))
    keep = []
    arr = nat.cram_slice_array(slices, keep)
    n = len(slices)
    bases = sum(len(t["seq"]) for s in slices for t in s["truth"]) + 4096
    bam, rec_off, st = engine.cram_decode_bam(arr, n, 3, 1, [], bases, bases * 3 + 400 * int(sum(s["nrec"] for s in slices)))
    assert (st == 0).all()
    h = engine.cram_batch_stage(arr, n, 3, 1, bases)
    try:
        runs = []
        for _ in range(3):
            d, nb, nr, nf, st2 = engine.cram_batch_decode_bam(h, n)
            assert (st2 == 0).all() and nr == sum(s["nrec"] for s in slices) and nf == n
            runs.append(bytes(engine.cram_batch_read_bam(h, nb)))
        assert runs[0] == runs[1] == runs[2] == bytes(bam)
    finally:
        engine.cram_batch_free(h)
    os.environ["HG_CRAM_RECORDS_PATH"] = "chain"
    try:
        bam_k, _, st_k = engine.cram_decode_bam(arr, n, 3, 1, [], bases, bases * 3 + 400 * int(sum(s["nrec"] for s in slices)))
    finally:
        del os.environ["HG_CRAM_RECORDS_PATH"]
    assert (st_k == 0).all() and bytes(bam_k) == bytes(bam)
    recs = T._parse_bam_records(bytes(bam))
    assert len(recs) == sum(s["nrec"] for s in slices)


@pytest.mark.gpu
def test_gpu_mixed_and_damaged_batches_get_the_chain_decoders_verdict(engine, hostlib):
    """fixture slices (CORE-coded: chain kernel) and synthetic ones (passes) in one call; then 150 damaged synthetic slices in one call: per-slice
    status and records equal the chain decoder compiled for the CPU"""
    from htslib_amd import synth_cram
    bound, dec = T._gpu_calls(engine)
    rng = np.random.default_rng(5)
    fx = [s for f, major, nref, s in T.load_slices() if f == "test/range.cram"]
    syn = [synth_cram.make_slice(rng, 300, 80), synth_cram.make_slice(rng, 90, 80, tags=True)]
    slices = [syn[0]] + fx[:1] + [syn[1]] + fx[1:]
    st, chain, _ = _raw(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, slices, 3, 7)
    st2, gpu, _ = _raw(bound, dec, slices, 3, 7)
    assert (st == 0).all() and (st2 == 0).all() and gpu == chain
    base = [synth_cram.make_slice(rng, 60, 70, tags=True), synth_cram.make_slice(rng, 45, 50, unmapped_every=4), synth_cram.make_slice(rng, 30, 64, detached_every=2, tags=True)]

    def mutate(b):
        b = bytearray(b)
        if not b: return bytes(b)
        k = int(rng.integers(0, 4))
        if k == 0:
            for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1: b = b[:int(rng.integers(0, len(b)))]
        elif k == 2: b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:
            i = int(rng.integers(0, len(b))); b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8))
        return bytes(b)

    bad = []
    while len(bad) < 150:
        s = dict(base[int(rng.integers(0, len(base)))])
        j = int(rng.integers(0, len(s["blocks"]))); bl = list(s["blocks"]); bl[j] = (bl[j][0], mutate(bl[j][1])); s["blocks"] = bl
        try:                                                             # keep what the CPU compile can render: the comparison below needs text on both sides
            stc, gc, _ = _raw(hostlib.hgr_host_records_bound, hostlib.hgr_host_decode_records, [s], 3, 1)
        except (AssertionError, ValueError, IndexError, UnicodeDecodeError, T.struct_error):
            continue
        bad.append((s, int(stc[0]), gc[0]))
    # one call with all of them.  The CPU compile saw every slice ALONE with a small seq_cap; in the batch the capacity is shared, so a verdict of
    # "does not fit" (-3) there may become 0 or -1 here -- everything else must agree, and no good slice may suffer from a damaged neighbour
    st_g, got_g, _ = _raw(bound, dec, [b[0] for b in bad], 3, 1)
    for k, b in enumerate(bad):
        if b[1] == 0: assert st_g[k] == 0 and got_g[k] == b[2], (k, st_g[k])
        elif b[1] == -1: assert st_g[k] == -1, (k, st_g[k])
        else: assert st_g[k] in (0, -1, -3), (k, st_g[k])
    assert sum(1 for b in bad if b[1] == 0) > 20 and sum(1 for b in bad if b[1] != 0) > 20
    # and one by one (same capacities as the CPU run): the verdicts are the chain decoder's, exactly
    for k, b in enumerate(bad[:60]):
        st1, got1, _ = _raw(bound, dec, [b[0]], 3, 1)
        assert int(st1[0]) == b[1], (k, st1, b[1])
        if b[1] == 0: assert got1[0] == b[2], k


@pytest.mark.gpu
def test_gpu_names_of_records_stored_without_one(engine):
    """files written without read names (RN = 0): cram_to_bam invents "<prefix>:<number of the record in the file>", the same for both mates of
    a pair (cram_decode.c:3113-3143); detached records keep their stored names.  Same stream from the passes and from the chain kernel."""
    from htslib_amd import _native as nat, synth_cram
    rng = np.random.default_rng(77)
    slices = [synth_cram.make_slice(rng, 400, 60, names=False, record_counter=100), synth_cram.make_slice(rng, 90, 60, names=False, record_counter=77, detached_every=3)]
    keep = []
    arr = nat.cram_slice_array(slices, keep)
    bases = sum(s["nrec"] for s in slices) * 60 + 4096
    out = {}
    for path in ("passes", "chain"):
        if path == "chain": os.environ["HG_CRAM_RECORDS_PATH"] = "chain"
        try:
            bam, rec_off, st = engine.cram_decode_bam(arr, len(slices), 3, 1, [], bases, bases * 4 + 800 * 500, name_prefix=b"in.cram")
        finally:
            os.environ.pop("HG_CRAM_RECORDS_PATH", None)
        assert (st == 0).all()
        out[path] = bytes(bam)
    assert out["passes"] == out["chain"]
    recs = T._parse_bam_records(out["passes"])
    k = 0
    for s, counter, det_every in zip(slices, (100, 77), (11, 3)):
        truth = s["truth"]
        has_name = [r % det_every == 0 for r in range(len(truth))]      # RN = 0: only detached records store one
        for r, t in enumerate(truth):
            name = recs[k][0][0]; k += 1
            mate = r + 1 if t["down"] else r - 1 if (r % 2 == 1 and truth[r - 1]["down"]) else None
            if has_name[r]: want = t["name"].decode()
            elif mate is not None and has_name[mate]: want = truth[mate]["name"].decode()      # "copy our mate if non-zero"
            else: want = "in.cram:%d" % (counter + (mate if mate is not None and mate < r else r) + 1)
            assert name == want, (r, name, want)
    assert k == len(recs)
    # without a prefix such records are called "*"
    bam, _, st = engine.cram_decode_bam(arr, len(slices), 3, 1, [], bases, bases * 4 + 800 * 500)
    names = [g[0] for g, _, _ in T._parse_bam_records(bytes(bam))]
    assert "*" in names
