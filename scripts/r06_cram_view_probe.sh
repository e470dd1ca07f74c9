#!/bin/bash
# GPU box: libhts-level CRAM tests, then the libhts_view CRAM legs of bench.py (64 and 512 slices)
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_libhts_gpu.py tests/test_cram_block_front.py tests/test_reference_cram.py -m gpu -x -q --timeout 900 2>&1 | tail -6
python - <<'PY'
import json, os, sys
sys.argv = ["bench.py", "--op", "e2e"]
sys.path.insert(0, os.getcwd())
import bench
class A: no_cpu_baseline = False
class R: ncores = os.cpu_count(); local = 0; args = A()
gpu, ref = os.path.join(bench.ROOT, "oracle", "_ref", "ref_view_gpu"), bench.REF_VIEW
for copies in (64,):
    print(json.dumps(bench.libhts_view_cram(R(), gpu, ref, [4, 8, 16, 64], copies=copies)), flush=True)
PY
