#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_libhts_gpu.py -x -q -k "cram31 or whole_slice or cram30 or htsjdk or transcode" > gpurun_out/r06_reader31_tests.txt 2>&1; tail -4 gpurun_out/r06_reader31_tests.txt | cut -c1-300
bash scripts/r06_cram31_libhts_probe.sh | grep -v "^$" | tail -16 | cut -c1-260
