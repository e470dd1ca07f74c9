#!/bin/bash
# GPU box: instruction-mix / stall counters of ONE kernel of a bench.py op (own pass: --pmc with --kernel-trace only).
#   bash scripts/pmc_kernel.sh <kernel-name-substring> <out-tag> <bench.py args...>     (env is passed through, e.g. HG_BENCH_RANS_FLAGS)
R=$GRAFT_REPO_ROOT; cd $R
K=$1; TAG=$2; shift 2
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r04
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"; do
  O=/tmp/pmc_$TAG; rm -rf $O; mkdir -p $O
  (cd /tmp && timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O -o pmc -- python $R/bench.py "$@" > $O/log.txt 2>&1)
  python3 - $O "$K" <<'PY' | tee -a $R/gpurun_out/r04/pmc_$TAG.txt
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if sys.argv[2] in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in agg.items()}, "(millions per dispatch, %d dispatches)" % max([len(v) for v in agg.values()] or [0]))
PY
done
