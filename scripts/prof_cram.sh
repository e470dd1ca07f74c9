#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_cram_records_fast.py tests/test_cram_records.py -m gpu -q 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_cram
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cram -o cram -- python $R/bench.py --op cram --steps 3 --no-cpu-baseline > $R/gpurun_out/prof_cram.log 2>&1
tail -1 $R/gpurun_out/prof_cram.log | cut -c1-600
python3 - $R/gpurun_out/prof_cram <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:18]:
        print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%9.1f ms' % (float(r['TotalDurationNs']) / 1e6), '%8.2f avg' % (float(r['AverageNs']) / 1e6), r['Percentage'])
PY
