#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python - <<'PY' > gpurun_out/r06_cram_reader_probe2.txt 2>&1
import json, os, sys, subprocess, time
sys.argv = ["bench.py", "--op", "e2e"]
sys.path.insert(0, os.getcwd())
import bench, numpy as np
from htslib_amd import _native as nat, synth_cram
eng = nat.Engine(0)
base = [synth_cram.make_slice(np.random.default_rng(7 + i), 10000, 150) for i in range(4)]
gpu = os.path.join(bench.ROOT, "oracle", "_ref", "ref_view_gpu"); ref = bench.REF_VIEW
w = bench.RefCramWorkload(eng, base, 64)
cram = os.path.join(w.dir, "in_l5.cram")
r = subprocess.run([ref, "-@", "32", "-C", "-o", "version=3.0", "-t", w.fa, "-p", cram, w.bam], capture_output=True)
for env in ({"HTS_GPU_STATS": "1"}, {"HTS_GPU_STATS": "1", "HG_CRAM_RECORDS_TIMING": "1"}):
    t = time.perf_counter()
    p = subprocess.run([gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, cram], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    print(round(time.perf_counter() - t, 3)); print(p.stderr.decode()[-6000:])
w.close()
PY
cat gpurun_out/r06_cram_reader_probe2.txt
