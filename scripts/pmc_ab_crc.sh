#!/bin/bash
# A/B of the BGZF inflate kernel's HBM traffic counters with and without its CRC pass (Weak 8): two library builds
# (product, and -DHG_AB_NO_CRC_PASS), FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes around tests/native/kbench.
#   bash scripts/pmc_ab_crc.sh <outdir> <file.bgzf>
OUT=$(realpath -m "$1"); F=$(realpath "$2"); ROOT=$(pwd)
mkdir -p "$OUT"; export TMPDIR=/tmp
for v in crc nocrc; do
  LIB=$ROOT/htslib_amd/libhtsgpu.so; [ $v = nocrc ] && LIB=$ROOT/variants/nocrc.so
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/${v}_$c" -o pmc -- $ROOT/tests/native/kbench $F 2 $LIB > "$OUT/${v}_$c.log" 2>&1)
  done
  (cd /tmp && $ROOT/tests/native/kbench $F 5 $LIB) > "$OUT/${v}_time.txt" 2>&1
done
python3 - "$OUT" <<'PY' > "$OUT/summary.json"
import csv, glob, collections, json, sys
res = {}
for v in ("crc", "nocrc"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for p in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (sys.argv[1], v, c), recursive=True):
            for r in csv.DictReader(open(p)):
                if "bgzf_inflate_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: vals.append(float(r["Counter_Value"]))
        res["%s_%s_KiB_per_dispatch" % (v, c)] = sum(vals) / len(vals) if vals else None
    res[v + "_time"] = open("%s/%s_time.txt" % (sys.argv[1], v)).read().strip().splitlines()[-1:]
print(json.dumps(res, indent=1))
PY
cat "$OUT/summary.json"
