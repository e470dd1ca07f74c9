#!/bin/bash
O=gpurun_out/enc; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_cram_encode.py tests/test_cram_records.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
HG_CRAM_RECORDS_TIMING=1 timeout 600 python bench.py --op encode --steps 5 > $O/bench_encode.json 2> $O/bench_encode.err; echo "encode rc=$?"; tail -4 $O/bench_encode.err; cut -c1-1500 $O/bench_encode.json
