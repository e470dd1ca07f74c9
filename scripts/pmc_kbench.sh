#!/bin/bash
# PMC passes around kbench (no python): bash scripts/pmc_kbench.sh <outdir> <file.bgzf> <lib.so>
OUT=$(realpath -m "$1"); F=$2; LIB=$(realpath "$3"); ROOT=$(pwd)
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 60 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -o pmc -- $ROOT/tests/native/kbench $F 1 $LIB > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$? : $grp" >> "$OUT/summary.txt"
done
python3 - "$OUT" <<'PY' >> "$OUT/summary.txt"
import csv, glob, collections, sys
for p in sorted(glob.glob(sys.argv[1] + '/p*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'inflate' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(p.split('/')[-2], {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
