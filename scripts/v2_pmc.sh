#!/bin/bash
# GPU box: instruction counters of the v2 kernels (separate PMC pass, kernel-trace only)
R=$GRAFT_REPO_ROOT; cd $R
python scripts/prep_bgzf.py 1 /dev/shm/k.bgzf >/dev/null
HG_INFLATE_V2=1 timeout 120 tests/native/kbench /dev/shm/k.bgzf 2 variants/prof.so 2>&1 | grep -v "in-kernel wave time"
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/pmc_v2
HG_INFLATE_V2=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_v2 -o pmc -- $R/tests/native/kbench /dev/shm/k.bgzf 1 $R/htslib_amd/libhtsgpu.so > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in glob.glob("$R/gpurun_out/pmc_v2/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
