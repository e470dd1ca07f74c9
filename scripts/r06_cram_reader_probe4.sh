#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python - <<'PY' > gpurun_out/r06_cram_reader_probe4.txt 2>&1
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
    for rep in range(2):
        t = time.perf_counter()
        p = subprocess.run(["strace", "-c", "-f", gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, cram] if False else [gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, cram], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, HTS_GPU_STATS="1"))
        print(copies * 4, "slices:", round(time.perf_counter() - t, 3)); print(p.stderr.decode()[-3000:])
    w.close()
PY
cat gpurun_out/r06_cram_reader_probe4.txt
