#!/bin/bash
# the whole-slice writer under cram_put_bam_seq: the libhts-level CRAM tests, then test_view -C on 256 and 1024 slices against stock
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_libhts_gpu.py -x -q -k "cram or htsjdk or whole_slice or index" > gpurun_out/r06_writer_tests.txt 2>&1
tail -25 gpurun_out/r06_writer_tests.txt
python - <<'PY' > gpurun_out/r06_cram_writer_probe.txt 2>&1
import json, os, sys, subprocess, time
sys.argv = ["bench.py", "--op", "e2e"]
sys.path.insert(0, os.getcwd())
import bench, numpy as np
from htslib_amd import _native as nat, synth_cram
eng = nat.Engine(0)
base = [synth_cram.make_slice(np.random.default_rng(7 + i), 10000, 150) for i in range(4)]
gpu = os.path.join(bench.ROOT, "oracle", "_ref", "ref_view_gpu"); ref = bench.REF_VIEW
for copies in (64, 256):
    w = bench.RefCramWorkload(eng, base, copies)
    def one(exe, th, env=None, out="/dev/null"):
        best = None
        for _ in range(2):
            t = time.perf_counter()
            p = subprocess.run([exe, "-@", str(th), "-C", "-o", "version=3.0", "-t", w.fa, "-p", out, w.bam], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t
            if p.returncode: return "rc %d %s" % (p.returncode, p.stderr.decode()[-300:])
            best = dt if best is None else min(best, dt)
        return round(best, 3)
    print("slices", copies * 4, "records", w.nrec, flush=True)
    print("  ours  whole-slice writer -@4", one(gpu, 4), flush=True)
    print("  ours  per-block path     -@64", one(gpu, 64, dict(os.environ, HTS_GPU_CRAM_SLICE="0")), flush=True)
    for th in (8, 16, 64): print("  stock -@%d" % th, one(ref, th), flush=True)
    out = os.path.join(w.dir, "o.cram")
    print("  ours to a file", one(gpu, 4, None, out), os.path.getsize(out), " stock to a file", one(ref, 16, None, out + "2"), os.path.getsize(out + "2"))
    a = subprocess.run([ref, "-@", "16", "-i", "reference=" + w.fa, out], stdout=subprocess.PIPE).stdout
    b = subprocess.run([ref, "-@", "16", "-i", "reference=" + w.fa, out + "2"], stdout=subprocess.PIPE).stdout
    print("  stock reads ours == stock reads its own:", a == b, len(a))
    p = subprocess.run([gpu, "-@", "4", "-C", "-o", "version=3.0", "-t", w.fa, "-p", "/dev/null", w.bam], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, HTS_GPU_STATS="1"))
    print("\n".join(l for l in p.stderr.decode().splitlines() if "cram writer" in l or "bam_to_cram" in l or "process" in l))
    w.close()
PY
cat gpurun_out/r06_cram_writer_probe.txt
