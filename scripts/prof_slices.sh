#!/bin/bash
# GPU box: per-kernel times of the whole-slice CRAM 3.1 encode / decode bench (kernel trace only).  bash scripts/prof_slices.sh [slices]
R=$GRAFT_REPO_ROOT; cd $R
S=${1:-256}
export TMPDIR=/tmp
O=/tmp/prof_slices; rm -rf $O; mkdir -p $O $R/gpurun_out/r04
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o sl -- python -c "
import sys; sys.path.insert(0, '$R'); sys.path.insert(0, '$R/scripts')
import bench_cram_slices; bench_cram_slices.main($S, reps=5)
" > $O/log.txt 2>&1)
tail -6 $O/log.txt
F=$(find $O -name "*kernel_stats.csv" | head -1)
cp $F $R/gpurun_out/r04/cram_slices_${S}_kernel_stats.csv
python3 - $F <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %5s total %9.2f ms avg %8.3f ms max %8.3f ms %5s%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
PY
