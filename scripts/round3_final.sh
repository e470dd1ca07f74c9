#!/bin/bash
# Last measurement pass of round 3 on the GPU box (after the deflate kernel work): the default bench line, rocprofv3 kernel stats of the deflate op,
# the whole GPU test suite.   usage: bash scripts/round3_final.sh <outdir under gpurun_out>
OUT=$(realpath -m "$1"); ROOT=$(pwd); mkdir -p "$OUT"
t0=$(date +%s); python bench.py > "$OUT/bench_all.json" 2> "$OUT/bench_all.err"; echo "bench rc=$? seconds=$(( $(date +%s) - t0 ))"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_deflate" -o deflate -- python "$ROOT/bench.py" --op deflate --gib 4 --no-cpu-baseline --steps 3 > "$OUT/stats_deflate.log" 2>&1; echo "prof deflate rc=$?"
cd "$ROOT"
timeout 600 python -m pytest tests -m gpu -q --timeout 500 2>&1 | tail -4 | tee "$OUT/pytest_gpu_tail.txt"
python3 - "$OUT" <<'PY'
import glob, sys, json
out = sys.argv[1]
for p in sorted(glob.glob(out + '/stats_*/**/*kernel_stats.csv', recursive=True)):
    print(p); print(open(p).read()[:700])
try:
    d = json.loads(open(out + '/bench_all.json').read().strip().splitlines()[-1])
    print(json.dumps({k: d[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'roofline')}))
    print(json.dumps(d.get('extra', {}).get('bgzf_deflate', {}))[:600])
    print(json.dumps(d.get('extra', {}).get('end_to_end', d.get('end_to_end', {})))[:900])
except Exception as e:
    print("bench parse:", e, open(out + '/bench_all.err').read()[-1500:])
PY
