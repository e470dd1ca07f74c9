#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "tok3 or cram31 or htscodecs or slice or cram_blocks or entropy or arith or ransnx16 or fqz or cram_encode" > gpurun_out/r06_tests_after_tok3.txt 2>&1
tail -4 gpurun_out/r06_tests_after_tok3.txt
python bench.py --op cram --no-cpu-baseline > gpurun_out/r06_bench_cram_after_tok3.json 2> gpurun_out/r06_bench_cram_after_tok3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_cram_after_tok3.json").read().strip().splitlines()[-1])
print(d.get("metric"), d.get("value"), d.get("ms_per_step"), {k: v for k, v in d.get("config", {}).items() if "GBps" in k or "ms" in k})
PY
