#!/bin/bash
# Round-4 measurement on the GPU box (every step bounded): rocprofv3 kernel stats (CSV) of the single ops, the slice bench's kernel timeline, C5 at its
# per-GPU size (1250 slices), HBM traffic counters of the deflate / inflate kernels through kbench, C1 round trip.
#   usage: bash scripts/round4_measure.sh <outdir under gpurun_out> [parts: stats slices c5 traffic c1 writer arith]
# (rocprofv3 writes CSV only with --output-format csv; every command that could wait on stdin reads /dev/null: a bare `head "$f"` with an empty $f once held a box for 15 minutes)
OUT=$(realpath -m "$1"); shift; PARTS="${*:-stats slices c5 traffic c1}"
ROOT=$GRAFT_REPO_ROOT; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
prof() { name=$1; shift; timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -o $name -- python "$ROOT/bench.py" "$@" > "$OUT/stats_$name.log" 2>&1; echo "prof $name rc=$?";
         f=$(find "$OUT/stats_$name" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv" && head -4 "$f" | cut -c1-200; rm -rf "$OUT/stats_$name"; }
for part in $PARTS; do case $part in
stats)
  prof inflate --op inflate --no-cpu-baseline --steps 5
  prof deflate --op deflate --gib 4 --no-cpu-baseline --steps 2
  prof rans32 --op rans --no-cpu-baseline --steps 10 --no-variants
  prof rans4 --op rans --nway 4 --no-cpu-baseline --steps 10
  prof records --op records --steps 5 --no-cpu-baseline
  prof encode --op encode --steps 5 --no-cpu-baseline
  prof fqz --op fqz --steps 5 --no-cpu-baseline ;;
slices)
  O=$OUT/tl; rm -rf $O; mkdir -p $O
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o sl -- python -c "
import sys; sys.path.insert(0, '$ROOT'); sys.path.insert(0, '$ROOT/scripts')
import bench_cram_slices; bench_cram_slices.main(256, reps=5)
" > $OUT/slices_256.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/cram_slices_256_kernel_stats.csv
  t=$(find $O -name "*kernel_trace.csv" | head -1); [ -n "$t" ] || t=/dev/null
  python3 - "$t" > $OUT/cram_slices_256_timeline.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
# calls are separated by gaps of more than 5 ms without any kernel
calls, cur, last_end = [], [], None
for s, e, n, q in ev:
    if last_end is not None and s - last_end > 5_000_000 and cur: calls.append(cur); cur = []
    cur.append((s, e, n, q)); last_end = e if last_end is None else max(last_end, e)
if cur: calls.append(cur)
print("%d kernel bursts (host calls); the long kernels (>= 2 ms) of each burst, start offset / duration in ms, queue:" % len(calls))
for i, c in enumerate(calls):
    t0 = c[0][0]; span = (max(e for _, e, _, _ in c) - t0) / 1e6
    print("burst %d: %d kernels, span %.1f ms" % (i, len(c), span))
    for s, e, n, q in c:
        if e - s >= 2_000_000: print("    +%7.1f  %7.1f  q%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
PY
  tail -40 $OUT/cram_slices_256_timeline.txt; rm -rf $O ;;
c5)
  cd $ROOT; timeout 500 python bench.py --op cram --slices 1250 --steps 5 > $OUT/bench_cram_1250.json 2> $OUT/bench_cram_1250.err; echo "c5 rc=$?"; cut -c1-700 $OUT/bench_cram_1250.json; cd /tmp ;;
traffic)
  cd $ROOT; bash scripts/pmc_traffic_kbench.sh 1 | tail -3; cp gpurun_out/r04/hbm_traffic_kbench.json $OUT/ 2>/dev/null; cd /tmp ;;
c1)
  cd $ROOT; timeout 500 python scripts/c1_bgzip_roundtrip.py 1 > "$OUT/c1_bgzip_roundtrip.json" 2> "$OUT/c1.err"; echo "c1 rc=$?"; cut -c1-600 "$OUT/c1_bgzip_roundtrip.json"; cd /tmp ;;
writer)   # the BAM -> CRAM 3.1 file writer: stage and round times (HTS_GPU_STATS) with the rANS + tok3 sets and with the range coder's, 64 and 1248 slices
  cd $ROOT
  for fl in 1 3; do for sl in 64 1248; do
    HG_BENCH_CRAM31_FLAGS=$fl HTS_GPU_STATS=1 timeout 400 python bench.py --op cram31 --slices $sl --steps 3 --no-cpu-baseline > "$OUT/cram31_f${fl}_s${sl}.json" 2> "$OUT/cram31_f${fl}_s${sl}.err" < /dev/null
    echo "writer flags $fl slices $sl rc=$?"; grep "hts-gpu" "$OUT/cram31_f${fl}_s${sl}.err" | tail -4; cut -c1-400 "$OUT/cram31_f${fl}_s${sl}.json"
  done; done; cd /tmp ;;
arith)    # the range coder's two encoder forms: few-stream calls, and the A/B on a batch of 256 slices (profiles/r04_arith_two_phase.txt)
  cd $ROOT
  timeout 60 python scripts/probe_arith_few.py > "$OUT/arith_few_two_phase.txt" 2>&1 < /dev/null; HG_ARITH_2P=0 timeout 60 python scripts/probe_arith_few.py > "$OUT/arith_few_one_pass.txt" 2>&1 < /dev/null
  paste "$OUT/arith_few_two_phase.txt" "$OUT/arith_few_one_pass.txt" | cut -c1-200
  for m in 8192 262144; do HG_ARITH_2P_MIN=$m timeout 120 python bench.py --op cram --slices 256 --steps 5 --no-cpu-baseline > "$OUT/cram_2pmin_$m.json" 2>/dev/null < /dev/null; echo "2P from $m:"; cut -c1-330 "$OUT/cram_2pmin_$m.json"; done
  cd /tmp ;;
esac; done
