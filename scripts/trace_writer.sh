#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_writer
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_writer -o tr -- python $R/scripts/write_timeline.py 2 > $R/gpurun_out/trace_writer.log 2>&1
grep "^rep" $R/gpurun_out/trace_writer.log
python3 - $R/gpurun_out/trace_writer <<'PY'
import csv, glob, sys
ev = []
for p in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-28:], r.get('Queue_Id', '?')))
for p in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'][12:], '-'))
ev.sort()
t0 = ev[0][0]
for s, e, k, q in ev[-90:-10]:
    if e - s > 5000: print(f"{(s - t0) / 1e6:10.3f} ms  +{(e - s) / 1e3:8.1f} us  {k:30s} q={q}")
PY
