"""Copies the .bai indexes that reference htslib wrote for two of its own BAM fixtures (tests/golden/bgzf holds the
BAMs and their plain images) -- the record boundaries and per-reference counts in them pin oracle/bam_oracle.c.
Run in the build container only (needs /root/reference)."""
import os, shutil
REF = "/root/reference/test"
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bam")
os.makedirs(HERE, exist_ok=True)
for f in ("colons.bam.bai", "range.bam.bai"):
    shutil.copy(os.path.join(REF, f), os.path.join(HERE, f))
print(sorted(os.listdir(HERE)))
