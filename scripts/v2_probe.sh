#!/bin/bash
# GPU box: parity tests of the two-kernel inflate + kbench (timing, in-kernel profile) + per-kernel times from rocprofv3
R=$GRAFT_REPO_ROOT; cd $R
HG_INFLATE_V2=1 timeout 300 python -m pytest tests/test_bgzf_inflate_gpu.py -m gpu -q -x --timeout 200 2>&1 | tail -8
python scripts/prep_bgzf.py ${1:-2} /dev/shm/k.bgzf >/dev/null
HG_INFLATE_V2=1 timeout 120 tests/native/kbench /dev/shm/k.bgzf 5 htslib_amd/libhtsgpu.so variants/prof.so 2>&1 | grep -v "in-kernel wave time"
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_v2
HG_INFLATE_V2=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v2 -o v2 -- $R/tests/native/kbench /dev/shm/k.bgzf 3 $R/htslib_amd/libhtsgpu.so > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/prof_v2/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "kernel" in r["Name"]: print(r["Name"].split("(")[0], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "max_us", round(float(r["MaxNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
