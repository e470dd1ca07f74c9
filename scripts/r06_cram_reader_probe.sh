#!/bin/bash
# the whole-slice reader under cram_get_bam_seq (cram_reader_front.c): the libhts-level tests, then test_view -B on CRAM 3.0 files of 256 and 1024 slices
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_libhts_gpu.py -x -q > gpurun_out/r06_reader_tests.txt 2>&1
tail -5 gpurun_out/r06_reader_tests.txt
python - <<'PY' > gpurun_out/r06_cram_reader_probe.txt 2>&1
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
    cram = os.path.join(w.dir, "in_l5.cram")
    r = subprocess.run([ref, "-@", "32", "-C", "-o", "version=3.0", "-t", w.fa, "-p", cram, w.bam], capture_output=True)
    print("slices", copies * 4, "records", w.nrec, "cram bytes", os.path.getsize(cram), flush=True)
    def one(exe, th, env=None, extra=()):
        best = None
        for _ in range(2):
            t = time.perf_counter()
            p = subprocess.run([exe, "-@", str(th), "-B", "-i", "reference=" + w.fa, *extra, cram], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t
            if p.returncode: return "rc %d %s" % (p.returncode, p.stderr.decode()[-300:])
            best = dt if best is None else min(best, dt)
        return round(best, 3)
    for th in (4, 16):
        print("  ours  whole-slice reader -@%d" % th, one(gpu, th), flush=True)
    print("  ours  per-block path     -@64", one(gpu, 64, dict(os.environ, HTS_GPU_CRAM_SLICE="0")), flush=True)
    for th in (8, 16, 64):
        print("  stock -@%d" % th, one(ref, th), flush=True)
    # CRAM -> BAM (what `samtools view -b in.cram` does)
    def conv(exe, th):
        t = time.perf_counter()
        p = subprocess.run([exe, "-@", str(th), "-b", "-i", "reference=" + w.fa, "-p", os.path.join(w.dir, "o.bam"), cram], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        return round(time.perf_counter() - t, 3) if p.returncode == 0 else p.stderr.decode()[-300:]
    print("  CRAM -> BAM ours -@4", conv(gpu, 4), " stock -@16", conv(ref, 16), " stock -@64", conv(ref, 64), flush=True)
    p = subprocess.run([gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, cram], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, HTS_GPU_STATS="1"))
    print(p.stderr.decode()[-1500:])
    w.close()
PY
cat gpurun_out/r06_cram_reader_probe.txt
