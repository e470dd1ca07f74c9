#!/bin/bash
# closing run of round 6: the whole -m gpu suite, smoke(), the default bench line
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_pytest_gpu_final.txt 2>&1; tail -3 gpurun_out/r06_pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); python bench.py > gpurun_out/r06_bench_default_line.json 2> gpurun_out/r06_bench_default.err; E=$(date +%s); echo "bench seconds $((E-S))"
cp gpurun_out/bench_full.json gpurun_out/r06_bench_default_full.json
