#!/bin/bash
# GPU box: where the libhts-level decode time goes -- start-up (N=1000 records) vs the whole 12 M records, ours vs stock at several thread counts
R=$GRAFT_REPO_ROOT; cd $R
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import bench
comp = bench.prepare(0x5EED0001, 4 << 30, 6, max(1, (os.cpu_count() or 2) - 4), None)
open("/dev/shm/v.bam", "wb").write(comp)
PY
t() { local s=$(date +%s.%N); "$@" > /dev/null 2>/tmp/err.txt; local e=$(date +%s.%N); echo "$(echo "$e - $s" | bc -l 2>/dev/null || python3 -c "print($e-$s)")"; }
for n in 1000 12000000; do
  for rep in 1 2; do
    echo "N=$n ours -@4: $(t oracle/_ref/ref_view_gpu -@4 -B -N $n /dev/shm/v.bam)"
  done
  for th in 4 8 16 64; do echo "N=$n stock -@$th: $(t oracle/_ref/ref_view -@$th -B -N $n /dev/shm/v.bam)"; done
done
echo "--- HTS_GPU_STATS"
HTS_GPU_STATS=1 oracle/_ref/ref_view_gpu -@4 -B -N 12000000 /dev/shm/v.bam 2>&1 >/dev/null | tail -15
