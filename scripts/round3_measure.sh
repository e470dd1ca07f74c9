#!/bin/bash
# Round-3 measurement on the GPU box: the default bench line (headline + extra), rocprofv3 kernel stats (CSV) of the single ops, HBM traffic PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only), the bgzip round trip.   usage: bash scripts/round3_measure.sh <outdir under gpurun_out>
OUT=$(realpath -m "$1"); ROOT=$(pwd); mkdir -p "$OUT"
t0=$(date +%s); python bench.py > "$OUT/bench_all.json" 2> "$OUT/bench_all.err"; echo "bench rc=$? seconds=$(( $(date +%s) - t0 ))"
cd /tmp; export TMPDIR=/tmp
prof() { name=$1; shift; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -o $name -- python "$ROOT/bench.py" "$@" > "$OUT/stats_$name.log" 2>&1; echo "prof $name rc=$?"; }
prof inflate --op inflate --no-cpu-baseline
prof records --op records --steps 5
prof encode --op encode --steps 5
prof rans --op rans --no-cpu-baseline --steps 10
prof deflate --op deflate --gib 4 --no-cpu-baseline --steps 2
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python "$ROOT/bench.py" --op inflate --no-cpu-baseline --steps 2 --warmup 0 > "$OUT/pmc_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o pmc -- python "$ROOT/bench.py" --op inflate --no-cpu-baseline --steps 2 --warmup 0 > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"; timeout 600 python scripts/c1_bgzip_roundtrip.py 1 > "$OUT/c1_bgzip_roundtrip.json" 2> "$OUT/c1.err"; echo "c1 rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in sorted(glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'inflate' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(p.split('/')[-3], {k: (sum(v) / len(v), len(v)) for k, v in agg.items()})
for p in sorted(glob.glob(out + '/stats_*/**/*kernel_stats.csv', recursive=True)):
    print(p); print(open(p).read()[:900])
PY
