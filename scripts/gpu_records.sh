#!/bin/bash
# GPU check of the CRAM record decoder: parity tests, bench line with phase times, rocprofv3 kernel stats
mkdir -p gpurun_out/records; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cram_records_fast.py tests/test_cram_records.py -m gpu -x -q > gpurun_out/records/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/records/pytest.log
tail -15 gpurun_out/records/pytest.log
HG_CRAM_RECORDS_TIMING=1 timeout 600 python bench.py --op records --steps 5 > gpurun_out/records/bench.json 2> gpurun_out/records/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/records/bench.err; cat gpurun_out/records/bench.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/records/prof -o rec -- python $R/bench.py --op records --steps 5 > $R/gpurun_out/records/prof.log 2>&1
find $R/gpurun_out/records/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -30 {}
