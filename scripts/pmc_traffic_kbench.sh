#!/bin/bash
# GPU box: HBM traffic counters (FETCH_SIZE, WRITE_SIZE: separate passes, --pmc with --kernel-trace only) of the BGZF deflate and inflate kernels
# through the native tests/native/kbench (a handful of dispatches: a python bench under --pmc serialises every torch kernel and takes minutes).
#   bash scripts/pmc_traffic_kbench.sh <GiB>      -> gpurun_out/r04/hbm_traffic_kbench.json
R=$GRAFT_REPO_ROOT; cd $R
G=${1:-1}
python scripts/prep_bgzf.py $G /dev/shm/k.bgzf >/dev/null
export TMPDIR=/tmp KBENCH_DEFLATE=1 KBENCH_LEVELS=6
mkdir -p $R/gpurun_out/r04
for c in FETCH_SIZE WRITE_SIZE; do
  O=/tmp/traf_$c; rm -rf $O; mkdir -p $O
  (cd /tmp && timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O -o pmc -- $R/tests/native/kbench /dev/shm/k.bgzf 1 $R/htslib_amd/libhtsgpu.so > $O/log.txt 2>&1)
  tail -3 $O/log.txt
done
python3 - $G <<'PY'
import csv, glob, collections, json, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob('/tmp/traf_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(p)):
            k = 'deflate' if 'deflate_kernel' in r['Kernel_Name'] else 'inflate' if 'bgzf_inflate_kernel' in r['Kernel_Name'] else None
            if k and r['Counter_Name'] == c: agg[k][c].append(float(r['Counter_Value']))
plain = os.path.getsize('/dev/shm/k.bgzf')
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) around tests/native/kbench on a %s GiB BAM (scripts/pmc_traffic_kbench.sh); counter unit KiB per dispatch, face value (gfx950: a wide coalesced read counts half, MI355X_MICROARCH.md)" % sys.argv[1],
       "compressed_file_bytes": plain}
for k, v in agg.items(): out[k] = {c: {"KiB_per_dispatch_mean": sum(x) / len(x), "dispatches": len(x)} for c, x in v.items()}
json.dump(out, open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/hbm_traffic_kbench.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
