#!/bin/bash
# the fused whole-slice reader: phase times of its runs on a 1024-slice CRAM 3.0 file, and the kernel table of the same command
R=$GRAFT_REPO_ROOT; cd $R
cd /tmp && export TMPDIR=/tmp; cd $R
python - <<'PY' > gpurun_out/r06_cram_reader_probe3.txt 2>&1
import json, os, sys, subprocess, time
sys.argv = ["bench.py", "--op", "e2e"]
sys.path.insert(0, os.getcwd())
import bench, numpy as np
from htslib_amd import _native as nat, synth_cram
eng = nat.Engine(0)
base = [synth_cram.make_slice(np.random.default_rng(7 + i), 10000, 150) for i in range(4)]
gpu = os.path.join(bench.ROOT, "oracle", "_ref", "ref_view_gpu"); ref = bench.REF_VIEW
w = bench.RefCramWorkload(eng, base, 256)
cram = os.path.join(w.dir, "in_l5.cram")
r = subprocess.run([ref, "-@", "32", "-C", "-o", "version=3.0", "-t", w.fa, "-p", cram, w.bam], capture_output=True)
for env in ({"HTS_GPU_STATS": "1"}, {"HTS_GPU_STATS": "1", "HG_CRAM_RECORDS_TIMING": "1"}):
    t = time.perf_counter()
    p = subprocess.run([gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, cram], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    print(round(time.perf_counter() - t, 3)); print(p.stderr.decode()[-6000:])
for th in (8, 16):
    t = time.perf_counter(); subprocess.run([ref, "-@", str(th), "-B", "-i", "reference=" + w.fa, cram], stdout=subprocess.DEVNULL); print("stock -@%d" % th, round(time.perf_counter() - t, 3))
os.makedirs("gpurun_out/r06_reader_prof", exist_ok=True)
p = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", "gpurun_out/r06_reader_prof", "-o", "reader", "--output-format", "csv", "--", gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, cram], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
print("rocprofv3 rc", p.returncode, p.stderr.decode()[-500:])
w.close()
PY
cat gpurun_out/r06_cram_reader_probe3.txt
find gpurun_out/r06_reader_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {}' 
find gpurun_out/r06_reader_prof -name "*kernel_trace.csv" -size +20M -delete
