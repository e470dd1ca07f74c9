#!/bin/bash
# GPU box: bgzf_read / bgzf_write / seek loops through libhts_bgzf.so with different numbers of window readers and copy helpers
cd $GRAFT_REPO_ROOT
for cfg in "1 3" "4 3" "4 7" "8 7" "8 15"; do
  set -- $cfg
  echo "== HTS_GPU_READ_THREADS=$1 HTS_GPU_COPY_THREADS=$2"
  HTS_GPU_READ_THREADS=$1 HTS_GPU_COPY_THREADS=$2 timeout 600 python bench.py --op e2e --gib 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['end_to_end']['gpu'])"
done
