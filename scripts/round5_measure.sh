#!/bin/bash
# Round-5 measurement on the GPU box: the commands behind profiles/r05_* (every step bounded by `timeout`, nothing reads stdin).
#   usage (from the repo root on the box): bash scripts/round5_measure.sh <outdir under gpurun_out> [parts: bench tests fqz fqzprof cram writer nx4 arith]
# rocprofv3 here writes the rocpd result database (--output-format rocpd; the CSV conversion of a --stats run of the slice bench took ten minutes);
# scripts/rocpd_stats.py turns a database into the kernel table and the burst timeline kept under profiles/.  Run from the repo root with TMPDIR=/tmp.
OUT=$(realpath -m "$1"); shift; PARTS="${*:-bench tests}"
mkdir -p "$OUT"; export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf "$OUT/prof_$name"; timeout 300 rocprofv3 --kernel-trace --output-format rocpd -d "$OUT/prof_$name" -o $name -- python bench.py "$@" > "$OUT/prof_$name.json" 2> "$OUT/prof_$name.err" < /dev/null
          db=$(find "$OUT/prof_$name" -name "*_results.db" | head -1); [ -n "$db" ] && python3 scripts/rocpd_stats.py "$db" --timeline 8 > "$OUT/${name}_kernel_stats.txt" 2>&1 < /dev/null; head -12 "$OUT/${name}_kernel_stats.txt" | cut -c1-150; }
for part in $PARTS; do case $part in
bench)     # the driver's line (profiles/r05_bench_default_line.json) and the uncompacted object (bench_full.json -> r05_bench_default_full.json)
  timeout 800 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" < /dev/null; echo "bench rc=$?"; cp bench_full.json "$OUT/bench_full.json"; tail -c 600 "$OUT/bench_default.json" ;;
tests)     # profiles/r05_pytest_gpu_full.log
  timeout 850 python -m pytest tests -q -m gpu -x 2>&1 < /dev/null | tail -15 > "$OUT/pytest_gpu.log"; cat "$OUT/pytest_gpu.log" ;;
fqz)       # profiles/r05_fqz_kernel_stats.txt + the A/B of the decoder's quality step
  trace fqz --op fqz --slices 1248 --steps 3 --no-cpu-baseline
  for q in 1 0; do echo "HG_FQZ_FAST=$q"; HG_FQZ_FAST=$q timeout 150 python bench.py --op fqz --slices 1248 --steps 3 --no-cpu-baseline 2>&1 < /dev/null | tail -1 | cut -c100-200; done ;;
fqzprof)   # profiles/r05_fqz_decode_phases.txt: needs `make fqzprof` (built here, travels with the snapshot)
  HTSGPU_LIB=$PWD/tests/native/libhtsgpu_fqzprof.so timeout 120 python bench.py --op fqz --slices 64 --steps 3 --no-cpu-baseline 2>&1 < /dev/null | grep "fqz-profile" | sort | uniq | sort -k3,3n | awk 'NR%4==1' ;;
cram)      # profiles/r05_cram_slices_256_final_timeline.txt + the per-round family times (HTS_GPU_STATS)
  trace cram --op cram --slices 256 --steps 3
  HTS_GPU_STATS=1 timeout 150 python bench.py --op cram --slices 256 --steps 3 > "$OUT/cram256_stats.log" 2>&1 < /dev/null; grep "auto-tuner round:" "$OUT/cram256_stats.log" | tail -12 | cut -c1-230 ;;
writer)    # profiles/r05_cram31_writer_rounds.txt: the 3.1 writer at C5's shape with and without the range coder's method sets
  for fl in 3 1; do HG_BENCH_CRAM31_FLAGS=$fl HTS_GPU_STATS=1 timeout 400 python bench.py --op cram31 --slices 1248 --steps 3 --no-cpu-baseline > "$OUT/cram31_f$fl.json" 2> "$OUT/cram31_f$fl.err" < /dev/null
    echo "writer flags $fl rc=$?"; grep "hts-gpu" "$OUT/cram31_f$fl.err" | tail -6 | cut -c1-260; cut -c1-400 "$OUT/cram31_f$fl.json"; done ;;
nx4)       # profiles/r05_nx4_encode.txt
  for v in 1 0; do echo "HG_NX4_SCALAR=$v"; HG_NX4_SCALAR=$v timeout 120 python scripts/probe_nx4_encode.py 2>&1 < /dev/null | tail -8; done ;;
arith)     # profiles/r05_arith_two_phase.txt
  timeout 120 python scripts/probe_arith2p.py 2>&1 < /dev/null | tail -20; HG_ARITH_2P_TIMES=1 timeout 120 python scripts/probe_arith2p_times.py 2>&1 < /dev/null | tail -20 ;;
esac; done
