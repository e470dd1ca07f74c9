import sys, os, zlib, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from htslib_amd import _native as nat, synth
from tests import refutil
eng = nat.Engine(0); orc = refutil.Oracle()
plain, _ = synth.bam_bgzf(1 << 20, level=6)
for name, data in [("bam", plain[:0xff00]), ("text", (b"the quick brown fox jumps over the lazy dog\n" * 2000)[:0xff00]), ("abc", b"abcdefgh" * 100), ("short", b"hello hello hello hello")]:
    comp = eng.bgzf_deflate_host(data, level=6, add_eof=False)
    open(os.path.join(ROOT, "gpurun_out", f"dbg_{name}.bgzf"), "wb").write(comp)
    open(os.path.join(ROOT, "gpurun_out", f"dbg_{name}.plain"), "wb").write(data)
    try:
        got = zlib.decompress(comp[18:-8], -15)
        d = next((i for i in range(min(len(got), len(data))) if got[i] != data[i]), None)
        print(name, len(data), len(comp), "decoded", len(got), "first diff", d)
    except Exception as e:
        print(name, len(data), len(comp), "zlib error", e)
