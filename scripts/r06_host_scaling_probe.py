"""How does the GPU box's host scale a codec port over processes that really run at the same time?  (bench.py's CPU baselines hand each of 254 workers a pickled 12-14 MB
sample through the pool's pipes: the workers start seconds apart.)  rANS Nx16 oracle decode of one 1.5 MB order-1 stream, nproc processes released by a barrier, 3 s each."""
import ctypes as C, multiprocessing as mp, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def worker(args):
    comp, bar, seconds = args
    comp = bytes(bytearray(comp))
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    orc.orc_ransnx16_uncompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    buf = C.create_string_buffer(1_500_064); got = C.c_size_t(0)
    orc.orc_ransnx16_uncompress(comp, len(comp), buf, 1_500_064, C.byref(got))
    bar.wait()
    t = time.perf_counter(); done = 0
    while time.perf_counter() - t < seconds:
        orc.orc_ransnx16_uncompress(comp, len(comp), buf, 1_500_064, C.byref(got)); done += got.value
    return done, time.perf_counter() - t
if __name__ == "__main__":
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(f): print(f, open(f).read().strip())
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    orc.orc_ransnx16_compress.restype = C.c_size_t; orc.orc_ransnx16_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_int]
    rng = np.random.default_rng(1)
    q = np.clip(np.cumsum(rng.integers(-2, 3, 1_500_000)) % 40 + 2, 2, 41).astype(np.uint8).tobytes()
    out = C.create_string_buffer(len(q) * 2 + 4096)
    n = orc.orc_ransnx16_compress(q, len(q), out, 1 | 4)
    comp = out.raw[:n]
    ctx = mp.get_context("fork")
    for nproc in (1, 16, 64, 128, 254):
        m = ctx.Manager(); bar = m.Barrier(nproc)
        with ctx.Pool(nproc) as pool:
            parts = pool.map(worker, [(comp, bar, 3.0)] * nproc, chunksize=1)
        print("nproc %3d: sum of rates %.2f GB/s, per process %.0f MB/s" % (nproc, sum(d / t for d, t in parts) / 1e9, sum(d / t for d, t in parts) / nproc / 1e6), flush=True)
        m.shutdown()
