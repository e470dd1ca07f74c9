#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_libhts_gpu.py -x -q -k "cram or htsjdk" > gpurun_out/r06_reader_tests2.txt 2>&1
tail -5 gpurun_out/r06_reader_tests2.txt
bash scripts/r06_cram_reader_probe2.sh
