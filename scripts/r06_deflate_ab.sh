#!/bin/bash
# GPU box: deflate parity tests, then kbench by level for libhtsgpu.so and the variants given (variants/*.so)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bgzf_deflate_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -8 | tee gpurun_out/deflate_tests.txt
python scripts/prep_bgzf.py ${GIB:-1} /dev/shm/k.bgzf
KBENCH_DEFLATE=1 KBENCH_LEVELS=${LEVELS:-156} timeout 600 tests/native/kbench /dev/shm/k.bgzf 2 htslib_amd/libhtsgpu.so "$@" 2>&1 | grep -v "in-kernel wave time\|v2 " | tee gpurun_out/deflate_ab.txt
