cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, subprocess, tempfile, sys
sys.path.insert(0, ".")
from tests import test_libhts_gpu as T
d = tempfile.mkdtemp(); T.unpack_fixtures(d)
env = T._env(d)
T.view(T.VIEW_REF, ["-t", "md.fa", "-S", "-C", "-o", "VERSION=3.0", "md#1.sam"], d, "stock.cram")
want = T.view(T.VIEW_REF, ["-D", "stock.cram"], d)
T.view(T.VIEW_GPU, ["-t", "md.fa", "-S", "-C", "-o", "VERSION=3.0", "md#1.sam"], d, "gpu.cram")
got = T.view(T.VIEW_REF, ["-D", "gpu.cram"], d)
for a, b in zip(want.decode().splitlines(), got.decode().splitlines()):
    if a != b: print("want", a); print("got ", b)
PY
