"""the block-method auto-tuner (cram_compress_block3 + struct cram_metrics, cram/cram_io.c:1912-2325).
The expectation comes from the REFERENCE's own function: tests/native/gen_metrics_ref.sh splices its text out of the
reference source into a scratch harness and drives it block by block with a cram_compress_by_method that returns scripted
sizes instead of compressing.  The engine's batch implementation (htslib_amd/csrc/cram_metrics_host.hip) gets the same
script through its test hook (no codec, no GPU needed) and must choose the same on-disk method and size for every block
and end with identical metrics -- whether the blocks arrive in one batch or in several."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import refutil

ROOT = refutil.ROOT
REF = "/root/reference"
M = lambda *ids: sum(1 << i for i in ids)
SETS = {"rans31": M(1, 5, 17, 18, 19, 20, 23, 12), "arith": M(1, 6, 25, 26, 27, 28, 29, 30, 31, 12), "v30": M(1, 11, 4, 16),
        "pack": M(1, 5, 17, 20, 21, 22, 23)}


def script(method, k, in_len):
    h = (method * 2654435761 + k * 40503 + 12345) & 0xffffffff
    h ^= h >> 15; h = (h * 2246822519) & 0xffffffff; h ^= h >> 13
    frac = 250 + h % 900
    if k % 4 == 2:
        frac = 1000 + h % 200
    return max(1, in_len * frac // 1000)


def block_len(i):
    n = 20000 + (i * 7919) % 5000 + (i % 4) * 30000
    return n * 20 if (400 <= i < 520 and i % 4 == 0) else n


class Metrics(C.Structure):
    _fields_ = [("trial", C.c_int), ("next_trial", C.c_int), ("consistency", C.c_int), ("sz", C.c_int * 32), ("input_avg_sz", C.c_int),
                ("input_avg_delta", C.c_int), ("method", C.c_int), ("revised_method", C.c_int), ("strat", C.c_int), ("cnt", C.c_int * 32),
                ("extra", C.c_double * 32), ("unpackable", C.c_int)]


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference source (its function is run, not copied)")
@pytest.mark.parametrize("level,version,setname,batch", [(5, 769, "rans31", 800), (5, 769, "rans31", 97), (1, 769, "rans31", 800), (7, 769, "arith", 250),
                                                         (9, 768, "v30", 800), (3, 1024, "pack", 800), (6, 1024, "pack", 31)])
def test_auto_tuner_equals_the_reference_function(built, tmp_path, level, version, setname, batch):
    n, mset = 800, SETS[setname]
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "native", "gen_metrics_ref.sh"), str(tmp_path), str(level), str(version), str(n), str(mset)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    want_b, want_m = [], {}
    for line in open(tmp_path / "ref_metrics.txt"):
        f = line.split()
        if f[0] == "B": want_b.append((int(f[2]), int(f[3])))
        elif f[0] == "M": want_m[int(f[1])] = f[2:]
    assert len(want_b) == n
    L = C.CDLL(os.path.join(ROOT, "htslib_amd", "libhtsgpu.so"))
    L.hg_cram_metrics_new.restype = C.POINTER(Metrics)
    base = [0]
    CB = C.CFUNCTYPE(C.c_uint32, C.c_int, C.c_size_t, C.c_uint32)
    cb = CB(lambda method, blk, in_len: script(method, base[0] + blk, in_len))
    L.hg_debug_set_cram_size_script(cb)
    ctr0 = (C.c_uint64 * 5)(); L.hg_debug_cram_tuner_counters(ctr0)
    try:
        mets = [L.hg_cram_metrics_new() for _ in range(4)]
        mets[3].contents.unpackable = 1
        got = []
        for b0 in range(0, n, batch):
            m = min(batch, n - b0)
            base[0] = b0
            lens = np.array([block_len(b0 + i) for i in range(m)], dtype=np.uint32)
            ins = [C.create_string_buffer(int(l)) for l in lens]
            outs = [C.create_string_buffer(2 * int(l) + 64) for l in lens]
            ip = (C.c_void_p * m)(*[C.addressof(x) for x in ins]); op = (C.c_void_p * m)(*[C.addressof(x) for x in outs])
            mp = (C.c_void_p * m)(*[C.addressof(mets[(b0 + i) % 4].contents) for i in range(m)])
            sets = np.full(m, mset, dtype=np.uint32); ol = np.zeros(m, dtype=np.uint32); used = np.zeros(m, dtype=np.int32)
            rc = L.hg_cram_compress_blocks_metrics_host(C.c_void_p(1), C.c_size_t(m), mp, sets.ctypes.data_as(C.c_void_p), level, version >> 8, ip,
                                                        lens.ctypes.data_as(C.c_void_p), op, ol.ctypes.data_as(C.c_void_p), used.ctypes.data_as(C.c_void_p))
            assert rc == 0
            got += [(int(used[i]), int(ol[i])) for i in range(m)]
        bad = [(i, got[i], want_b[i]) for i in range(n) if got[i] != want_b[i]]
        assert not bad, bad[:10]
        for q in range(4):
            mm = mets[q].contents
            ours = [mm.trial, mm.next_trial, mm.consistency, mm.input_avg_sz, mm.input_avg_delta, mm.method, mm.revised_method, mm.strat, mm.unpackable]
            ref = want_m[q]
            assert ours == [int(x) for x in ref[:9]], (q, ours, ref[:9])
            for k in range(32):
                assert (mm.sz[k], mm.cnt[k]) == (int(ref[9 + 3 * k]), int(ref[10 + 3 * k])) and abs(mm.extra[k] - float(ref[11 + 3 * k])) < 1e-5, (q, k)
        # trial phases ahead of time (cram_metrics_host.hip): in one 800-block call the later phases' trial blocks are compressed with the first round and folded from
        # the cache when the state machine reaches them -- the call takes a few rounds instead of one per phase, with the reference's decisions (checked above)
        ctr = (C.c_uint64 * 5)(); L.hg_debug_cram_tuner_counters(ctr)
        calls, rounds, ahead, folded, normal = [int(ctr[i] - ctr0[i]) for i in range(5)]
        print("auto-tuner: %d calls, %d rounds, %d trial blocks ahead of time (%d folded from the cache), %d trial blocks on the normal path" % (calls, rounds, ahead, folded, normal))
        if batch == 800 and os.environ.get("HG_CRAM_SPECULATE") != "0":
            assert folded > 0 and rounds <= 6, (rounds, ahead, folded, normal)
    finally:
        L.hg_debug_set_cram_size_script(None)
