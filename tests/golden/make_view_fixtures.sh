#!/bin/bash
# Makes tests/golden/view_fixtures.tar.gz: the reference's TEST DATA (no code) that test/test.pl's test_view and test_index subs run over
# (test/test.pl:708-830, 1067-1160) -- every test/*#*.sam with its FASTA, the pre-made htsjdk CRAMs, index*.sam / index.vcf and the golden
# indexes.  Consumed by tests/test_libhts_gpu.py on the GPU box, where /root/reference does not exist.  Needs /root/reference.
set -e
R=${REF:-/root/reference}/test; T=$(mktemp -d); HERE=$(cd "$(dirname "$0")" && pwd)
cd "$R"
cp *#*.sam auxf.fa* c1.fa* c2.fa* ce.fa* md.fa* xx.fa* auxf#values_java.cram ce#5b_java.cram range.cram xx#large_aux_java.cram \
   index.sam index_dos.sam index.vcf index2.sam index3.sam index3_exp.sam index.bam.bai index.bam.csi index.bcf.csi index.cram.crai \
   index.sam.gz.bai index.sam.gz.csi index.vcf.gz.csi index.vcf.gz.tbi range.bam range.cram.crai "$T"/
tar czf "$HERE/view_fixtures.tar.gz" --owner=0 --group=0 --mtime='2020-01-01' -C "$T" .
rm -rf "$T"
