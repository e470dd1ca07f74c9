#!/bin/bash
# GPU box: BGZF deflate kernel -- kbench timing + instruction-mix / utilisation PMC passes.   bash scripts/deflate_pmc.sh <GiB> lib.so ...
R=$GRAFT_REPO_ROOT; cd $R
G=$1; shift
python scripts/prep_bgzf.py $G /dev/shm/k.bgzf >/dev/null
export TMPDIR=/tmp KBENCH_DEFLATE=1 KBENCH_LEVELS=${KBENCH_LEVELS:-6}
for l in "$@"; do
  timeout 300 tests/native/kbench /dev/shm/k.bgzf 2 $R/$l 2>&1 | grep -i "deflate\|match detail"
  n=$(basename $l .so)
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    O=$R/gpurun_out/defl_pmc_$n; rm -rf $O; mkdir -p $O
    (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O -o pmc -- $R/tests/native/kbench /dev/shm/k.bgzf 1 $R/$l > $O/log.txt 2>&1)
    python3 - $O $n <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'deflate_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 1) for k, v in agg.items()}, "(millions per dispatch)")
PY
  done
done
