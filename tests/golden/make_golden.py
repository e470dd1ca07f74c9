#!/usr/bin/env python3
"""Freeze the reference's own BGZF known-answer vectors.

Run in the build container (needs /root/reference and oracle/_ref built by
`make -C oracle ref`).  For every BGZF-framed fixture under
/root/reference/test it stores
    tests/golden/bgzf/<flat name>          the compressed stream, byte for byte
    tests/golden/bgzf/<flat name>.plain    what the REAL reference decodes it to
                                           (oracle/_ref/ref_bgzip -dc, zlib build;
                                           cross-checked with the libdeflate build)
plus MANIFEST.json (sizes, block counts, deflate block types seen, md5 of plain).
The GPU box has no /root/reference; tests only read this directory.
"""
import hashlib, json, os, subprocess, sys, glob

REF = "/root/reference/test"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "bgzf")
BGZIP = os.path.join(HERE, "..", "..", "oracle", "_ref", "ref_bgzip")
BGZIP_LD = BGZIP + "_ld"
MAX_SIZE = 400_000   # keep the repo small


def btypes(data: bytes):
    """first deflate BTYPE of every block (0 stored, 1 fixed, 2 dynamic)."""
    pos, out = 0, []
    while pos < len(data):
        bs = (data[pos + 16] | (data[pos + 17] << 8)) + 1
        out.append((data[pos + 18] >> 1) & 3)
        pos += bs
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    man = {}
    for f in sorted(glob.glob(REF + "/**/*", recursive=True)):
        if not os.path.isfile(f) or os.path.getsize(f) > MAX_SIZE:
            continue
        d = open(f, "rb").read()
        if len(d) < 28 or d[:4] != b"\x1f\x8b\x08\x04" or d[12:14] != b"BC":
            continue
        plain = subprocess.run([BGZIP, "-dc", f], capture_output=True)
        plain_ld = subprocess.run([BGZIP_LD, "-dc", f], capture_output=True)
        if plain.returncode != 0 or plain_ld.returncode != 0 or plain.stdout != plain_ld.stdout:
            print("skip (reference refuses or flavours differ):", f, file=sys.stderr)
            continue
        name = os.path.relpath(f, REF).replace("/", "__")
        open(os.path.join(OUT, name), "wb").write(d)
        open(os.path.join(OUT, name + ".plain"), "wb").write(plain.stdout)
        bt = btypes(d)
        man[name] = {"source": "test/" + os.path.relpath(f, REF), "csize": len(d), "usize": len(plain.stdout),
                     "blocks": len(bt), "first_btype_hist": [bt.count(0), bt.count(1), bt.count(2)],
                     "md5_plain": hashlib.md5(plain.stdout).hexdigest()}
    json.dump(man, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(len(man), "fixtures,", sum(v["csize"] + v["usize"] for v in man.values()), "bytes")


if __name__ == "__main__":
    main()
