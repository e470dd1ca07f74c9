"""CRAM 3.1 write/read side over whole slices (BASELINE config C5 shape, one GPU): every data series of S slices goes
through hg_cram_compress_blocks_metrics_host (the reference's cram_compress_block2 loop) and back through
hg_cram_uncompress_blocks_host.  Host entry points => the rate includes PCIe both ways.  Pointer tables are built
once, outside the timed region."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from htslib_amd import _native as nat
from tests.test_rans4x8 import synth_series
from tests.test_tok3 import illumina_names



def main(S=64, device=0, reps=5, quiet=False):
    say = (lambda *a, **k: None) if quiet else print
    eng = nat.Engine(device)
    rng = np.random.default_rng(7)
    M = lambda *ids: sum(1 << i for i in ids)
    RANS = M(1, 5, 17, 18, 19, 20, 23)                       # GZIP + RANS_PR0/1/64/9/128/193 (cram_encode.c:818-826, level 5)
    ARITH = M(1, 6, 25, 26, 27, 28, 31)                      # the "small"/"archive" profile's range-coder sets
    series = {  # name: (generator, method set)
        "QS": (lambda: synth_series(rng, "qual4", 1_500_000), RANS), "BA": (lambda: synth_series(rng, "bases", 1_500_000), RANS),
        "RN": (lambda: illumina_names(rng, 10_000), M(1, 8)), "AP": (lambda: rng.integers(0, 300, 20_000, dtype=np.uint16).tobytes(), RANS),
        "BF": (lambda: rng.choice(np.array([99, 147, 83, 163], dtype=np.uint16), 10_000).tobytes(), RANS),
        "TS": (lambda: rng.integers(-600, 600, 10_000, dtype=np.int16).tobytes(), ARITH),
        "MQ": (lambda: rng.choice(np.array([0, 60], dtype=np.uint8), 10_000, p=[0.05, 0.95]).tobytes(), RANS),
        "NP": (lambda: rng.integers(0, 100_000_000, 10_000, dtype=np.uint32).tobytes(), ARITH),
    }
    proto = {k: [g() for _ in range(4)] for k, (g, _) in series.items()}
    metrics = {k: nat.lib.hg_cram_metrics_new() for k in series}
    datas, mets, sets = [], [], []
    for s in range(S):
        for k, (_, ms) in series.items():
            datas.append(proto[k][s % 4]); mets.append(metrics[k]); sets.append(ms)
    n = len(datas)
    plain = sum(map(len, datas))
    ins = [(C.c_char * max(len(d), 1)).from_buffer_copy(d) for d in datas]
    outs = [C.create_string_buffer(nat.lib.hg_cram_compress_bound(len(d))) for d in datas]
    ip = (C.c_void_p * n)(*[C.addressof(x) for x in ins]); op = (C.c_void_p * n)(*[C.addressof(x) for x in outs])
    il = np.array([len(d) for d in datas], dtype=np.uint32); ol = np.zeros(n, dtype=np.uint32)
    mk = np.array(sets, dtype=np.uint32); used = np.zeros(n, dtype=np.int32)
    mp = (C.c_void_p * n)(*mets)


    def encode():
        rc = nat.lib.hg_cram_compress_blocks_metrics_host(eng._h, n, mp, mk.ctypes.data, 5, 3, ip, il.ctypes.data, op, ol.ctypes.data, used.ctypes.data)
        assert rc == 0, rc


    t = time.perf_counter(); encode(); t0 = time.perf_counter() - t
    say("call 1 (trial phases inside): %.1f ms, %.2f GB/s" % (t0 * 1e3, plain / t0 / 1e9), flush=True)
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); encode(); ts.append(time.perf_counter() - t)
    enc_best = sorted(ts)[len(ts) // 2]                       # median of the steady calls
    say("steady calls: %s ms -> %.2f GB/s plain (%d slices, %d blocks, %.1f MB), ratio %.3f" % (
        ["%.1f" % (x * 1e3) for x in ts], plain / min(ts) / 1e9, S, n, plain / 1e6, ol.sum() / plain), flush=True)
    say("methods per series:", {k: int(used[i]) for i, k in enumerate(series)}, flush=True)
    # ---- read side
    comp = [outs[i].raw[:int(ol[i])] for i in range(n)]
    cin = [(C.c_char * max(len(c), 1)).from_buffer_copy(c if c else b"\0") for c in comp]
    dout = [C.create_string_buffer(max(len(d), 1)) for d in datas]
    cp = (C.c_void_p * n)(*[C.addressof(x) for x in cin]); dp = (C.c_void_p * n)(*[C.addressof(x) for x in dout])
    cl = np.array([len(c) for c in comp], dtype=np.uint32); st = np.zeros(n, dtype=np.int32); meth = used.astype(np.int32)
    ts = []
    for _ in range(max(5, reps)):
        t = time.perf_counter()
        rc = nat.lib.hg_cram_uncompress_blocks_host(eng._h, n, meth.ctypes.data, cp, cl.ctypes.data, dp, il.ctypes.data, st.ctypes.data)
        ts.append(time.perf_counter() - t)
        assert rc == 0 and (st == 0).all()
    assert all(dout[i].raw[:len(datas[i])] == datas[i] for i in range(n))
    say("decode calls: %s ms -> %.2f GB/s plain, verified" % (["%.1f" % (x * 1e3) for x in ts], plain / min(ts) / 1e9), flush=True)

    nser = len(series)
    return {"slices": S, "blocks": n, "plain_bytes": int(plain), "comp_bytes": int(ol.sum()), "encode_s": enc_best, "decode_s": sorted(ts)[len(ts) // 2],
            "methods": {k: int(used[i]) for i, k in enumerate(series)},
            "sample": [(int(used[i]), comp[i], datas[i]) for i in range(min(n, 4 * nser))]}       # (on-disk method, stream, plaintext) of four slices: the CPU baseline's input


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
