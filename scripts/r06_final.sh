#!/bin/bash
# the round's closing measurements: the default bench line (with CPU baselines), C1, the libhts BAM decode start-up probe
R=$GRAFT_REPO_ROOT; cd $R
S=$(date +%s); python bench.py > gpurun_out/r06_bench_default_line.json 2> gpurun_out/r06_bench_default.err; E=$(date +%s); echo "bench seconds $((E-S))" | tee gpurun_out/r06_bench_default_seconds.txt
cp gpurun_out/bench_full.json gpurun_out/r06_bench_default_full.json
timeout 600 python scripts/c1_bgzip_roundtrip.py 1 > gpurun_out/r06_c1_bgzip_roundtrip.json 2> gpurun_out/c1.err; echo "c1 rc=$?"
bash scripts/r06_view_probe.sh > gpurun_out/r06_view_probe_final.txt 2>&1; tail -16 gpurun_out/r06_view_probe_final.txt
