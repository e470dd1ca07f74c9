"""GPU box: device-resident throughput of the CRAM ITF8 series kernels (hg_cram_itf8_decode_dev / _encode_dev).
usage: bench_itf8.py [streams] [values per stream]"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from htslib_amd import _native as nat
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
eng = nat.Engine(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
DESC = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_len", "<u4"), ("scratch_off", "<u4"), ("reserved", "<u4")])
for label, gen in (("mostly 1-byte values (flags, lengths, AP deltas)", lambda: np.minimum(rng.geometric(0.02, nv), 3000).astype(np.int32)),
                   ("mixed 1..5-byte values", lambda: (rng.integers(0, 2**31, nv, dtype=np.int64) >> rng.integers(0, 31, nv)).astype(np.int32))):
    cols = [gen() for _ in range(64)]
    cols = [cols[i % 64] for i in range(ns)]
    vals = np.concatenate(cols)
    d_vals = torch.from_numpy(vals).to(dev)
    edesc = np.zeros(ns, dtype=DESC)
    edesc["in_off"] = np.arange(ns, dtype=np.uint64) * nv; edesc["in_len"] = nv
    edesc["out_off"] = np.arange(ns, dtype=np.uint64) * (5 * nv + 16); edesc["out_len"] = 5 * nv + 16
    d_edesc = torch.from_numpy(edesc.view(np.uint8)).to(dev)
    d_bytes = torch.zeros(ns * (5 * nv + 16), dtype=torch.uint8, device=dev)
    d_len = torch.zeros(ns, dtype=torch.int32, device=dev); d_st = torch.zeros(ns, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    enc = nat.lib.hg_cram_itf8_encode_dev; dec = nat.lib.hg_cram_itf8_decode_dev
    enc.argtypes = dec.argtypes = [C.c_void_p] * 4 + [C.c_size_t] + [C.c_void_p] * 4 if False else None
    enc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    dec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    def run_enc(): assert enc(eng._h, d_vals.data_ptr(), d_edesc.data_ptr(), ns, d_bytes.data_ptr(), d_len.data_ptr(), d_st.data_ptr(), s) == 0
    run_enc(); torch.cuda.synchronize()
    lens = d_len.cpu().numpy().astype(np.int64)
    assert int(d_st.abs().sum()) == 0
    ddesc = np.zeros(ns, dtype=DESC)
    ddesc["in_off"] = edesc["out_off"]; ddesc["in_len"] = lens; ddesc["out_off"] = np.arange(ns, dtype=np.uint64) * nv; ddesc["out_len"] = nv
    d_ddesc = torch.from_numpy(ddesc.view(np.uint8)).to(dev)
    d_out = torch.zeros(ns * nv, dtype=torch.int32, device=dev); d_cnt = torch.zeros(ns, dtype=torch.int32, device=dev)
    def run_dec(): assert dec(eng._h, d_bytes.data_ptr(), d_ddesc.data_ptr(), ns, d_out.data_ptr(), d_cnt.data_ptr(), d_st.data_ptr(), s) == 0
    run_dec(); torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0 and bool((d_out == d_vals).all())
    for name, fn in (("decode", run_dec), ("encode", run_enc)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
        nbytes = int(lens.sum())
        print(f"{label}: {name} {ns} streams x {nv} values, {nbytes / 1e6:.1f} MB of ITF8 <-> {4 * ns * nv / 1e6:.1f} MB of int32: {best:.3f} ms = "
              f"{ns * nv / best / 1e6:.2f} G values/s, {(nbytes + 4 * ns * nv) / best / 1e6:.1f} GB/s of algorithmic traffic")
