#!/bin/bash
O=gpurun_out/tests; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
