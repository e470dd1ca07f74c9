#!/bin/bash
# CRAM 3.1 -- the reference's DEFAULT output version since 1.22 -- at the libhts level: test_view on libhts_gpu.so (whole-slice writer / reader, Nx16 + tok3 on the device) against the
# same program on the reference's libhts, whose 3.1 codecs in this container are oracle/'s scalar restatements (ORC_STUB_CODECS31=1: htscodecs is absent) -- read the stock side as a floor
R=$GRAFT_REPO_ROOT; cd $R
python - <<'PY' > gpurun_out/r06_cram31_libhts_probe.txt 2>&1
import json, os, sys, subprocess, time
sys.argv = ["bench.py", "--op", "e2e"]
sys.path.insert(0, os.getcwd())
import bench, numpy as np
from htslib_amd import _native as nat, synth_cram
eng = nat.Engine(0)
base = [synth_cram.make_slice(np.random.default_rng(7 + i), 10000, 150) for i in range(4)]
gpu = os.path.join(bench.ROOT, "oracle", "_ref", "ref_view_gpu"); ref = bench.REF_VIEW
e31 = dict(os.environ, ORC_STUB_CODECS31="1")
for copies in (64, 256):
    w = bench.RefCramWorkload(eng, base, copies)
    ours = os.path.join(w.dir, "ours31.cram"); stock = os.path.join(w.dir, "stock31.cram")
    def t(cmd, env=None, reps=2):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter(); p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env); dt = time.perf_counter() - t0
            if p.returncode: return "rc %d %s" % (p.returncode, p.stderr.decode()[-200:])
            best = dt if best is None else min(best, dt)
        return round(best, 3)
    print("slices", copies * 4, "records", w.nrec, flush=True)
    print("  encode 3.1 (default version): ours -@4", t([gpu, "-@", "4", "-C", "-t", w.fa, "-p", ours, w.bam]), os.path.getsize(ours),
          "| stock -@64 on the scalar codecs", t([ref, "-@", "64", "-C", "-t", w.fa, "-p", stock, w.bam], e31, 1), os.path.getsize(stock), flush=True)
    print("  decode ours' file: ours -@4", t([gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, ours]), "| stock -@64 on the scalar codecs", t([ref, "-@", "64", "-B", "-i", "reference=" + w.fa, ours], e31, 1), flush=True)
    print("  decode stock's file: ours -@4", t([gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, stock]), "| stock -@64", t([ref, "-@", "64", "-B", "-i", "reference=" + w.fa, stock], e31, 1), flush=True)
    a = subprocess.run([ref, "-@", "64", "-i", "reference=" + w.fa, ours], stdout=subprocess.PIPE, env=e31).stdout
    b = subprocess.run([gpu, "-@", "4", "-i", "reference=" + w.fa, stock], stdout=subprocess.PIPE).stdout
    c = subprocess.run([ref, "-@", "64", "-i", "reference=" + w.fa, stock], stdout=subprocess.PIPE, env=e31).stdout
    import hashlib
    def canon(x): return hashlib.md5(b"\n".join(b"\t".join(f[:11] + sorted(f[11:])) if len(f) > 11 else ln for ln in x.split(b"\n") for f in [ln.split(b"\t")])).hexdigest()
    print("  the CPU restatements read ours == stock's own:", canon(a) == canon(c), " we read stock's == stock:", canon(b) == canon(c), len(c), flush=True)
    p = subprocess.run([gpu, "-@", "4", "-B", "-i", "reference=" + w.fa, ours], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, HTS_GPU_STATS="1"))
    print("\n".join(l for l in p.stderr.decode().splitlines() if "cram run" in l or "cram reader" in l))
    w.close()
PY
cat gpurun_out/r06_cram31_libhts_probe.txt
