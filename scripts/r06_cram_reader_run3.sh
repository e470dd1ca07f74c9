#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash scripts/r06_cram_reader_probe3.sh > gpurun_out/r06_probe3_console.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu -k "cram" > gpurun_out/r06_reader_tests3.txt 2>&1
tail -5 gpurun_out/r06_reader_tests3.txt
head -60 gpurun_out/r06_probe3_console.txt
