#!/bin/bash
# GPU box: CRAM slice bench (scripts/bench_cram_slices.py N) alone and under rocprofv3 kernel stats
R=$GRAFT_REPO_ROOT; N=${1:-256}
cd $R; [ -z "$SKIP_PLAIN" ] && timeout 600 python scripts/bench_cram_slices.py $N 2>&1 | tail -25
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_cram
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cram -o cram -- env PYTHONPATH=$R python $R/scripts/bench_cram_slices.py $N > $R/gpurun_out/prof_cram.log 2>&1; tail -3 $R/gpurun_out/prof_cram.log
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/prof_cram/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(r["Name"].split("(")[0][:70], "calls", r["Calls"], "total_ms", round(float(r["TotalDurationNs"]) / 1e6, 1), "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
