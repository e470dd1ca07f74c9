#!/bin/bash
# round 3, GPU pass A: slice-compress test, rANS bench (variants, all-stream verification, cpu baseline) + rocprofv3 stats, PMC A/B of the CRC pass, fqz / cram ops
O=gpurun_out/r3a; mkdir -p $O; cd $GRAFT_REPO_ROOT; R=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cram_block_front.py -m gpu -x -q -k "compress_slice" > $O/pytest_slice.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_slice.log
timeout 900 python bench.py --op rans --steps 10 > $O/bench_rans.json 2> $O/bench_rans.err; echo "rans rc=$?"; cut -c1-1500 $O/bench_rans.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_rans -o rans -- python $R/bench.py --op rans --steps 10 --no-cpu-baseline --slices 1000 > $R/$O/prof_rans.log 2>&1)
python scripts/rocpd_to_csv.py $(find $O/prof_rans -name "*.db" | head -1) > $O/r03_rans_kernel_stats.csv; head -5 $O/r03_rans_kernel_stats.csv | cut -c1-200
python scripts/prep_bgzf.py 1 /tmp/ab.bgzf > $O/prep.log 2>&1
bash scripts/pmc_ab_crc.sh $O/pmc_ab /tmp/ab.bgzf > $O/pmc_ab.log 2>&1; cat $O/pmc_ab/summary.json
timeout 600 python bench.py --op fqz --steps 5 > $O/bench_fqz.json 2> $O/bench_fqz.err; echo "fqz rc=$?"; cut -c1-1200 $O/bench_fqz.json
timeout 900 python bench.py --op cram --steps 5 > $O/bench_cram.json 2> $O/bench_cram.err; echo "cram rc=$?"; cut -c1-2000 $O/bench_cram.json
