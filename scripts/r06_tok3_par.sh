#!/bin/bash
# the position-major tok3 name kernel: parity tests, then the kernel table of 256 blocks x 10 000 names with it and with the serial walk (HG_TOK3_PAR=0)
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_tok3.py tests/test_htscodecs_front.py -m gpu -x -q > gpurun_out/r06_tok3_tests.txt 2>&1
tail -15 gpurun_out/r06_tok3_tests.txt
cat > /tmp/tk.py <<'PY'
import sys, time
import numpy as np
sys.path.insert(0, ".")
from htslib_amd import _native as nat
from tests.test_tok3 import illumina_names
eng = nat.Engine(0)
names = [illumina_names(np.random.default_rng(i), 10_000, paired=(i & 1) == 1) for i in range(8)]
blocks = [names[i % 8] for i in range(256)]
enc = eng.tok3_encode_host(blocks, [0] * 256)
for _ in range(4):
    t = time.perf_counter(); out = eng.cram_uncompress_blocks([(8, e, len(d)) for e, d in zip(enc, blocks)]); dt = time.perf_counter() - t
assert all(o == b for o, b in zip(out[0], blocks)) and not out[1].any()
print("256 tok3 blocks, %.1f MB of names: %.1f ms per call" % (sum(map(len, blocks)) / 1e6, dt * 1e3))
PY
cd /tmp && export TMPDIR=/tmp && cd $R
for par in 1 0; do
  HG_TOK3_PAR=$par timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06_tok3_prof_$par -o t --output-format csv -- python /tmp/tk.py 2>/dev/null | grep "tok3 blocks"
  echo "HG_TOK3_PAR=$par"; f=$(find gpurun_out/r06_tok3_prof_$par -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
  find gpurun_out/r06_tok3_prof_$par -name "*kernel_trace.csv" -delete
done
