#!/usr/bin/env python3
"""Freeze every block of the reference's CRAM v3.0 fixtures as a vector for the CRAM block layer
(cram_uncompress_block, cram/cram_io.c:1576-1754): on-disk method id, compressed payload, declared
uncompressed size, and the expected plaintext where it can be produced WITHOUT our code:
RAW = the payload, GZIP = Python's zlib (independent inflate), rANS 4x8 QS blocks = QUAL-33 from
the .sam twin (see make_golden_rans.py).  Needs /root/reference."""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_rans as R

OUT = os.path.join(R.HERE, "cram_blocks.json")


def main():
    rman = json.load(open(os.path.join(R.OUT, "MANIFEST.json")))
    rans_expected = {}
    for k, v in rman.items():
        rans_expected[(v["source"], v["content_id"], v["csize"], v["usize"])] = v["expected_hex"]
    out = []
    import glob
    extra = sorted(os.path.relpath(f, R.REF) for f in glob.glob(os.path.join(R.REF, "tlen", "*.cram")))   # written by htslib itself
    for cram in [c for c, _ in R.PAIRS] + extra:
        b = open(os.path.join(R.REF, cram), "rb").read()
        for nrec, blks in R.containers(b):
            for blk in blks:
                (method, ctype, cid, csz, usz, data) = blk
                exp = None
                if method == 0: exp = data
                elif method == 1: exp = zlib.decompress(data, 15 + 32)
                elif method == 4: 
                    h = rans_expected.get(("test/" + cram, cid, csz, usz))
                    exp = bytes.fromhex(h) if h else None
                if exp is not None: assert len(exp) == usz, (cram, method, cid)
                out.append({"source": "test/" + cram, "method": method, "content_type": ctype, "content_id": cid,
                            "usize": usz, "data_hex": data.hex(), "expected_hex": exp.hex() if exp is not None else None,
                            "hdr_hex": blk.hdr.hex(), "crc32": blk.crc})
    json.dump(out, open(OUT, "w"))
    from collections import Counter
    print(len(out), "blocks", Counter(o["method"] for o in out), "with plaintext:", sum(o["expected_hex"] is not None for o in out))


if __name__ == "__main__":
    main()
