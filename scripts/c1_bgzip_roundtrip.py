#!/usr/bin/env python3
"""BASELINE config 1: `bgzip -c` / `bgzip -d` round trip of a 1 GiB synthetic FASTQ -- with the reference's OWN bgzip.c
linked to our library (oracle/_ref/bgzip_gpu, threads flag = batch mode) and, beside it, stock htslib (oracle/_ref/ref_bgzip_ld,
libdeflate, all host cores).  Files live in /dev/shm.   usage: c1_bgzip_roundtrip.py [GiB] > json"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from htslib_amd import synth

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
shm = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
src = os.path.join(shm, "c1.fastq")
n = int(gib * (1 << 30))
with open(src, "wb") as f:
    piece = synth.fastq(64 << 20)
    done = 0
    while done < n:
        w = piece[:min(len(piece), n - done)]
        f.write(w); done += len(w)
ncores = os.cpu_count() or 1


def timed(cmd, out):
    t = time.perf_counter()
    with open(out, "wb") as o:
        r = subprocess.run(cmd, stdout=o, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t
    assert r.returncode == 0, (cmd, r.stderr[-500:])
    return dt


res = {"config": "bgzip -c / -d round trip of a %.2f GiB synthetic FASTQ, files in /dev/shm" % gib, "plain_bytes": n}
for name, exe, at in (("gpu", os.path.join(ROOT, "oracle", "_ref", "bgzip_gpu"), "-@4"), ("reference", os.path.join(ROOT, "oracle", "_ref", "ref_bgzip_ld"), f"-@{ncores}")):
    if not os.path.exists(exe):
        continue
    gz, back = os.path.join(shm, f"c1_{name}.gz"), os.path.join(shm, f"c1_{name}.out")
    tc = min(timed([exe, at, "-c", src], gz) for _ in range(2))
    td = min(timed([exe, at, "-d", "-c", gz], back) for _ in range(2))
    ok = subprocess.run(["cmp", "-s", src, back]).returncode == 0
    # fixed cost of a run: the same program on a one-block file (process start, and for the GPU build the HIP runtime + context + first pinned buffers)
    tiny, tiny_gz = os.path.join(shm, f"c1_{name}_tiny"), os.path.join(shm, f"c1_{name}_tiny.gz")
    open(tiny, "wb").write(open(src, "rb").read(40000))
    fixed_c = min(timed([exe, at, "-c", tiny], tiny_gz) for _ in range(3))
    fixed_d = min(timed([exe, at, "-d", "-c", tiny_gz], tiny + ".out") for _ in range(3))
    for f in (tiny, tiny_gz, tiny + ".out"):
        os.unlink(f)
    res[name] = {"program": os.path.relpath(exe, ROOT) + " " + at, "compress_GBps": round(n / tc / 1e9, 3), "decompress_GBps": round(n / td / 1e9, 3),
                 "compress_seconds": round(tc, 3), "decompress_seconds": round(td, 3),
                 "one_block_file_seconds": {"compress": round(fixed_c, 3), "decompress": round(fixed_d, 3)},
                 "compress_GBps_net_of_fixed": round(n / max(tc - fixed_c, 1e-6) / 1e9, 3),
                 "decompress_GBps_net_of_fixed": round(n / max(td - fixed_d, 1e-6) / 1e9, 3),
                 "compressed_bytes": os.path.getsize(gz), "round_trip_identical": ok}
    os.unlink(back)
# cross decodes
if "gpu" in res and "reference" in res:
    a = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_bgzip_ld"), f"-@{ncores}", "-d", "-c", os.path.join(shm, "c1_gpu.gz")], stdout=open(os.path.join(shm, "c1_x.out"), "wb"))
    res["stock_htslib_decodes_our_file"] = a.returncode == 0 and subprocess.run(["cmp", "-s", src, os.path.join(shm, "c1_x.out")]).returncode == 0
    b = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "bgzip_gpu"), "-@4", "-d", "-c", os.path.join(shm, "c1_reference.gz")], stdout=open(os.path.join(shm, "c1_x.out"), "wb"))
    res["we_decode_stock_htslib_file"] = b.returncode == 0 and subprocess.run(["cmp", "-s", src, os.path.join(shm, "c1_x.out")]).returncode == 0
    res["size_vs_reference_libdeflate6"] = round(res["gpu"]["compressed_bytes"] / res["reference"]["compressed_bytes"], 4)
for f in ("c1.fastq", "c1_gpu.gz", "c1_reference.gz", "c1_x.out"):
    try: os.unlink(os.path.join(shm, f))
    except OSError: pass
print(json.dumps(res))
