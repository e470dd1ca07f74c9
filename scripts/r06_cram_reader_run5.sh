#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_libhts_gpu.py -x -q -k "whole_slice or cram30 or htsjdk" > gpurun_out/r06_reader_tests5.txt 2>&1; tail -3 gpurun_out/r06_reader_tests5.txt
bash scripts/r06_cram_reader_probe4.sh | grep "slices:\|cram run\|cram reader\|process"
