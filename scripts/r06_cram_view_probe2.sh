#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python - <<'PY'
import json, os, sys, subprocess, time
sys.argv = ["bench.py", "--op", "e2e"]
sys.path.insert(0, os.getcwd())
import bench, numpy as np
from htslib_amd import _native as nat, synth_cram
eng = nat.Engine(0)
base = [synth_cram.make_slice(np.random.default_rng(7 + i), 10000, 150) for i in range(4)]
w = bench.RefCramWorkload(eng, base, 16)
cram = os.path.join(w.dir, "in_l5.cram")
subprocess.run([bench.REF_VIEW, "-C", "-o", "version=3.0", "-t", w.fa, "-p", cram, w.bam], check=True)
gpu = os.path.join(bench.ROOT, "oracle", "_ref", "ref_view_gpu")
env = dict(os.environ, HTS_GPU_STATS="1")
t = time.time(); p = subprocess.run([gpu, "-@", "16", "-B", "-i", "reference=" + w.fa, cram], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env); print("decode", time.time() - t)
print(p.stderr.decode()[-3500:])
w.close()
PY
