#!/bin/bash
# Round measurement on the GPU box: the default bench line (headline + extra), rocprofv3 kernel stats of the same command,
# HBM traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only).
# usage: bash scripts/round_measure.sh <outdir under gpurun_out>
OUT=$(realpath -m "$1"); ROOT=$(pwd); mkdir -p "$OUT"
python bench.py > "$OUT/bench_all.json" 2> "$OUT/bench_all.err"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o inflate -- python "$ROOT/bench.py" --op inflate --no-cpu-baseline > "$OUT/stats.log" 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python "$ROOT/bench.py" --op inflate --no-cpu-baseline --steps 2 --warmup 0 > "$OUT/pmc_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- python "$ROOT/bench.py" --op inflate --no-cpu-baseline --steps 2 --warmup 0 > "$OUT/pmc_write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_deflate" -o deflate -- python "$ROOT/bench.py" --op deflate --gib 4 --no-cpu-baseline --steps 2 > "$OUT/stats_deflate.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in sorted(glob.glob(out + '/pmc_*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'inflate' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(p.split('/')[-2], {k: (sum(v) / len(v), len(v)) for k, v in agg.items()})
for p in glob.glob(out + '/stats*/*kernel_stats.csv'):
    print(p); print(open(p).read()[:1200])
PY
