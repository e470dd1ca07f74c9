#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 1 4; do timeout 900 python scripts/c1_bgzip_roundtrip.py $g > gpurun_out/c1_${g}g.json 2> gpurun_out/c1_${g}g.err; echo "c1 $g rc=$?"; cat gpurun_out/c1_${g}g.json; echo; done
