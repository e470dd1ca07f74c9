"""write a synthetic BAM (BGZF, zlib-6) of <gib> GiB plain with blocks of at most <blk> bytes to <out> (occupancy experiments of the deflate kernel)."""
import os, sys, zlib, struct
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htslib_amd import synth
gib, blk, out = float(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
def part(i):
    r = synth.bam_stream(64 << 20, seed=0x5EED0001 + i, with_header=(i == 0))
    plain = bytes(r[0] if isinstance(r, tuple) else r)
    return b"".join(synth.bgzf_block(plain[o:o + blk], 6) for o in range(0, len(plain), blk))
n = max(1, int(gib * 16))
with ProcessPoolExecutor(max(1, (os.cpu_count() or 2) - 4)) as ex:
    parts = list(ex.map(part, range(n)))
open(out, "wb").write(b"".join(parts) + synth.BGZF_EOF)
print("prepared", sum(map(len, parts)))
