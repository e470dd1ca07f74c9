#!/bin/bash
# GPU box: occupancy experiment of the deflate kernel: a file of <= 32000-byte blocks through the product library (2 workgroups per CU) and a build
# with half the staged input (3 per CU); then the usual A/B on full-size blocks
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python scripts/prep_bgzf_small.py 1 32000 /dev/shm/ks.bgzf
KBENCH_DEFLATE=1 KBENCH_LEVELS=156 timeout 600 tests/native/kbench /dev/shm/ks.bgzf 2 htslib_amd/libhtsgpu.so variants/occ3.so 2>&1 | grep -v "in-kernel wave time\|v2 " | tee gpurun_out/deflate_occ.txt
python scripts/prep_bgzf.py 1 /dev/shm/k.bgzf
KBENCH_DEFLATE=1 KBENCH_LEVELS=156 timeout 600 tests/native/kbench /dev/shm/k.bgzf 2 htslib_amd/libhtsgpu.so variants/wide12.so variants/g116.so 2>&1 | grep -v "in-kernel wave time\|v2 " | tee gpurun_out/deflate_ab.txt
