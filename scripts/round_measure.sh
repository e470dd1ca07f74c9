#!/bin/bash
# Round measurement on the GPU box: headline bench, rocprofv3 kernel stats, HBM traffic PMC passes.
# usage: bash scripts/round_measure.sh <outdir under gpurun_out>
OUT=$(realpath -m "$1"); ROOT=$(pwd); mkdir -p "$OUT"
python bench.py > "$OUT/bench_inflate.json" 2> "$OUT/bench_inflate.err"
python bench.py --op deflate --gib 4 > "$OUT/bench_deflate.json" 2> "$OUT/bench_deflate.err"
python bench.py --op rans > "$OUT/bench_rans.json" 2> "$OUT/bench_rans.err"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o inflate -- python "$ROOT/bench.py" --no-cpu-baseline > "$OUT/stats.log" 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python "$ROOT/bench.py" --no-cpu-baseline --steps 2 --warmup 0 > "$OUT/pmc_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- python "$ROOT/bench.py" --no-cpu-baseline --steps 2 --warmup 0 > "$OUT/pmc_write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in sorted(glob.glob(out + '/pmc_*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'inflate' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(p.split('/')[-2], {k: (sum(v) / len(v), len(v)) for k, v in agg.items()})
for p in glob.glob(out + '/stats/*kernel_stats.csv'):
    print(open(p).read()[:1500])
PY
