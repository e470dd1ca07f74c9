#!/bin/bash
# GPU box: bgzf_read / bgzf_write / seek loops through libhts_bgzf.so + the reference programs on the library (drop-in check after front-end changes)
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --op e2e --gib 6 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['end_to_end']['first_handle_ms'], d['end_to_end']['gpu'])"
python scripts/read_timeline.py 6 | tail -2
timeout 900 python -m pytest tests/test_bgzf_front_gpu.py tests/test_reference_programs.py -m gpu -q 2>&1 | tail -3
