#!/bin/bash
# GPU box: deflate tests + kbench A/B (scripts/deflate_ab.sh), then the PMC passes of scripts/deflate_pmc.sh for the product library at level 6
R=$GRAFT_REPO_ROOT; cd $R
scripts/deflate_ab.sh "$@"
KBENCH_LEVELS=6 bash scripts/deflate_pmc.sh 1 htslib_amd/libhtsgpu.so 2>&1 | tee gpurun_out/deflate_pmc.txt
