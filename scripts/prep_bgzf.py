"""write a synthetic BAM (BGZF, zlib-6) of <gib> GiB plain to <out> using all host cores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
gib, out = float(sys.argv[1]), sys.argv[2]
t = time.time()
comp = bench.prepare(0x5EED0001, int(gib * (1 << 30)), 6, max(1, (os.cpu_count() or 2) - 4), None)
open(out, "wb").write(comp)
print(f"prepared {len(comp)} bytes in {time.time()-t:.1f}s")
