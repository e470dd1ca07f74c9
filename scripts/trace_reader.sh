#!/bin/bash
# GPU box: kernel + memory-copy trace of a bgzf_read loop (are the pipes' jobs overlapping?)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_reader
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_reader -o tr -- python $R/scripts/read_timeline.py 2 > $R/gpurun_out/trace_reader.log 2>&1
tail -2 $R/gpurun_out/trace_reader.log
python3 - $R/gpurun_out/trace_reader <<'PY'
import csv, glob, sys
ev = []
for p in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'inflate' in r['Kernel_Name']: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r.get('Queue_Id', r.get('Stream_Id', '?'))))
for p in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'][:12], '-'))
ev.sort()
t0 = ev[0][0]
# last 40 events of the run (steady state)
for s, e, k, q in ev[-48:]:
    print(f"{(s - t0) / 1e6:10.3f} ms  +{(e - s) / 1e3:8.1f} us  {k:14s} q={q}")
PY
