#!/bin/bash
# GPU box: round 6's rocprofv3 evidence -- kernel stats of the default bench's inflate + deflate legs, HBM traffic counters of the deflate / inflate kernels
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $R/bench.py --op deflate --gib 10 --steps 5 --warmup 1 --no-cpu-baseline > /tmp/ks_deflate.log 2>&1)
python3 - <<'PY'
import csv, glob
for p in glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(p)))
    out = open('gpurun_out/r06/r06_deflate_kernel_stats.txt', 'w')
    out.write("rocprofv3 --kernel-trace --stats -- python bench.py --op deflate --gib 10 --steps 5 --warmup 1 --no-cpu-baseline\n")
    for r in rows[:12]:
        out.write("%-70s calls %6s  total %12s ns  avg %12s ns  %6s %%\n" % (r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']))
    out.close()
    print(open('gpurun_out/r06/r06_deflate_kernel_stats.txt').read())
PY
tail -2 /tmp/ks_deflate.log | head -c 600; echo
rm -rf /tmp/ks
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $R/bench.py --op inflate --gib 10 --steps 5 --warmup 1 --no-cpu-baseline > /tmp/ks_inflate.log 2>&1)
python3 - <<'PY'
import csv, glob
for p in glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(p)))
    out = open('gpurun_out/r06/r06_inflate_kernel_stats.txt', 'w')
    out.write("rocprofv3 --kernel-trace --stats -- python bench.py --op inflate --gib 10 --steps 5 --warmup 1 --no-cpu-baseline\n")
    for r in rows[:8]:
        out.write("%-70s calls %6s  total %12s ns  avg %12s ns  %6s %%\n" % (r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']))
    out.close()
    print(open('gpurun_out/r06/r06_inflate_kernel_stats.txt').read())
PY
tail -1 /tmp/ks_inflate.log | head -c 400; echo
sed -i 's#gpurun_out/r04#gpurun_out/r06#g' scripts/pmc_traffic_kbench.sh
bash scripts/pmc_traffic_kbench.sh 1 2>&1 | tail -4
