#!/bin/bash
# PMC passes for the inflate kernel (each counter group in its own run; FETCH_SIZE / WRITE_SIZE separate).
# usage (on the GPU box, from the repo root): bash scripts/pmc_inflate.sh <outdir> [bench args...]
set -u
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 2 --warmup 0 $*"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pass$i" -o pmc -- python "$ROOT/bench.py" $ARGS > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$? : $grp" >> "$OUT/summary.txt"
done
find "$OUT" -name "*.csv" | head -40 >> "$OUT/summary.txt"
