#!/bin/bash
# GPU box: the whole -m gpu suite, then the headline op alone (inflate, 10 GiB, CPU baseline skipped) for a quick number.
O=gpurun_out/tests; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py --op inflate --no-cpu-baseline > gpurun_out/bench_inflate_quick.json 2> gpurun_out/bench_inflate_quick.err; echo "bench rc=$?"; cat gpurun_out/bench_inflate_quick.json
