"""GPU box: throughput probe of the fqzcomp quality codec (CRAM method 7) through the host entry points (PCIe included).
usage: bench_fqz.py [streams] [records per stream] [read length]
The oracle (one CPU core) is timed on one stream beside it."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from htslib_amd import _native as nat
from tests import refutil
from tests.test_fqzcomp import reads
eng = nat.Engine(0)
rng = np.random.default_rng(1)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NREC = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
RLEN = int(sys.argv[3]) if len(sys.argv) > 3 else 150
orc = refutil.FqzOracle()
base, recs = [], []
for k in range(4):
    q, lens, fl = reads(rng, NREC, RLEN, True, 41)
    base.append((q, orc.encode(q, lens, fl, k % 4, 0)))
    recs.append((lens, fl))
t = time.perf_counter(); orc.encode(base[0][0], recs[0][0], recs[0][1], 0, 6); te = time.perf_counter() - t
print("oracle encode (1 core): %.1f MB/s" % (len(base[0][0]) / te / 1e6), flush=True)
t = time.perf_counter(); rc, back, _ = orc.decode(base[0][1], len(base[0][0])); tc = time.perf_counter() - t
assert rc == 0 and back == base[0][0]
print("oracle decode (1 core): %.1f MB/s; stream ratio %.3f" % (len(back) / tc / 1e6, len(base[0][1]) / len(back)), flush=True)
for ns in sorted({min(NS, 64), NS}):
    blocks = [(7, base[i % 4][1], len(base[i % 4][0])) for i in range(ns)]
    qb = sum(b[2] for b in blocks)
    eng.cram_uncompress_blocks(blocks[:2])
    ts = []
    for _ in range(2):
        t = time.perf_counter(); outs, st = eng.cram_uncompress_blocks(blocks); ts.append(time.perf_counter() - t)
    assert (st == 0).all() and outs[0] == base[0][0] and outs[ns - 1] == base[(ns - 1) % 4][0]
    print("fqz decode %5d streams x %d B: %8.3f GB/s (host API, best of 2: %.1f ms; %.2f MB/s per stream)"
          % (ns, blocks[0][2], qb / min(ts) / 1e9, min(ts) * 1e3, blocks[0][2] / min(ts) / 1e6), flush=True)
    datas = [base[i % 4][0] for i in range(ns)]
    args = (datas, [recs[i % 4][0] for i in range(ns)], [recs[i % 4][1] for i in range(ns)], [i % 4 for i in range(ns)])
    ts = []
    for _ in range(2):
        t = time.perf_counter(); enc = eng.fqz_encode_host(*args); ts.append(time.perf_counter() - t)
    back, st = eng.cram_uncompress_blocks([(7, e, len(d)) for e, d in zip(enc[:8], datas[:8])])
    assert (st == 0).all() and back[0] == datas[0]
    print("fqz encode %5d streams x %d B: %8.3f GB/s (host API, best of 2: %.1f ms; ratio %.3f)"
          % (ns, len(datas[0]), qb / min(ts) / 1e9, min(ts) * 1e3, sum(map(len, enc)) / qb), flush=True)
