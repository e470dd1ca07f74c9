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
gpu = os.path.join(bench.ROOT, "oracle", "_ref", "ref_view_gpu")
for th in (4, 16, 64):
    for wb in ("1", "0"):
        env = dict(os.environ, HTS_GPU_CRAM_WRITEBEHIND=wb)
        t = time.time(); p = subprocess.run([gpu, "-@", str(th), "-C", "-o", "version=3.0", "-t", w.fa, "-p", "/dev/null", w.bam], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        print("ours encode -@%d writebehind=%s" % (th, wb), round(time.time() - t, 3), p.returncode, flush=True)
env = dict(os.environ, HTS_GPU_STATS="1")
p = subprocess.run([gpu, "-@", "16", "-C", "-o", "version=3.0", "-t", w.fa, "-p", "/dev/null", w.bam], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
print(p.stderr.decode()[-6000:])
w.close()
PY
