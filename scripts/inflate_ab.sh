#!/bin/bash
# GPU box: A/B of BGZF inflate kernel builds with tests/native/kbench (status of every block = CRC check) + instruction-mix PMC pass.
#   bash scripts/inflate_ab.sh <GiB> lib1.so lib2.so ...
R=$GRAFT_REPO_ROOT; cd $R
G=$1; shift
python scripts/prep_bgzf.py $G /dev/shm/k.bgzf >/dev/null
LIBS=""; for l in "$@"; do LIBS="$LIBS $R/$l"; done
timeout 300 tests/native/kbench /dev/shm/k.bgzf 5 $LIBS 2>&1 
export TMPDIR=/tmp
for l in "$@"; do
  n=$(basename $l .so); O=$R/gpurun_out/ab_pmc_$n; rm -rf $O; mkdir -p $O
  (cd /tmp && timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O -o pmc -- $R/tests/native/kbench /dev/shm/k.bgzf 1 $R/$l > $O/log.txt 2>&1)
  python3 - $O $n <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'bgzf_inflate_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 1) for k, v in agg.items()}, "(millions per dispatch)")
PY
done
