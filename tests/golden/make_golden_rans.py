#!/usr/bin/env python3
"""Freeze rANS 4x8 known-answer vectors from the reference's CRAM v3.0 fixtures.

htscodecs (the reference implementation of the CRAM entropy codecs) is an absent submodule, so the
expected plaintext cannot come from running the reference.  It is derived INDEPENDENTLY of any
rANS decoder from the fixture's .sam / .bam twin: the QS data series of a CRAM slice is the concatenation
of the records' quality values (QUAL - 33), the RN series the read names each followed by the
BYTE_ARRAY_STOP byte, BF / RL / AP the ITF8-coded flags, read lengths and positions (absolute, or deltas
from the slice start when the preservation map says AP is delta-coded), TS the ITF8-coded template lengths,
SC the soft-clipped bases of each S CIGAR op followed by the stop byte, a one-byte aux tag (type c / C) the
tag values of the records in order, and a string tag (type Z) ITF8(length + 1), the value and its NUL per
record (the 900 KB ZZ:Z tag of xx#large_aux is the largest real order-1 stream the fixtures hold).  Blocks whose content cannot be derived that way are stored with
their declared raw size only ("size-only" vectors).

Output: tests/golden/rans4x8/<file>.<n>.bin (compressed block payload), MANIFEST.json
(content id, order, raw size, series name, expected plaintext hex or null; plaintexts over 4 KiB go to
<file>.<n>.bin.plain.z, zlib-packed, named by "expected_z").
Needs /root/reference; run in the build container.
"""
import json, os, re, struct, sys, zlib

REF = "/root/reference/test"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "rans4x8")
PAIRS = [("ce#5b_java.cram", "ce#5b.sam"), ("auxf#values_java.cram", "auxf#values.sam"),
         ("xx#large_aux_java.cram", "xx#large_aux.sam"), ("range.cram", "range.bam")]


def itf8(b, p):
    v = b[p]
    if v < 0x80: return v, p + 1
    if v < 0xC0: return ((v & 0x3F) << 8) | b[p + 1], p + 2
    if v < 0xE0: return ((v & 0x1F) << 16) | (b[p + 1] << 8) | b[p + 2], p + 3
    if v < 0xF0: return ((v & 0x0F) << 24) | (b[p + 1] << 16) | (b[p + 2] << 8) | b[p + 3], p + 4
    return ((v & 0x0F) << 28) | (b[p + 1] << 20) | (b[p + 2] << 12) | (b[p + 3] << 4) | (b[p + 4] & 0x0F), p + 5


def ltf8(b, p):
    v = b[p]; n = 0
    while n < 8 and (v & (0x80 >> n)): n += 1
    if n == 0: return v, p + 1
    val = v & (0xFF >> (n + 1)) if n < 8 else 0
    for i in range(n): val = (val << 8) | b[p + 1 + i]
    return val, p + 1 + n


def put_itf8(v):
    v &= 0xFFFFFFFF
    if v < 0x80: return bytes([v])
    if v < 0x4000: return bytes([0x80 | (v >> 8), v & 0xFF])
    if v < 0x200000: return bytes([0xC0 | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000: return bytes([0xE0 | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([0xF0 | (v >> 28), (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])


def parse_preservation(d):
    """-> {key: first value byte} of the preservation map (RN, AP, RR are one-byte booleans)"""
    p = 0
    sz, p = itf8(d, p); end = p + sz
    n, p = itf8(d, p)
    out = {}
    for _ in range(n):
        key = bytes(d[p:p + 2]).decode(); p += 2
        if key in ("RN", "AP", "RR"): out[key] = d[p]; p += 1
        elif key == "SM": p += 5
        elif key == "TD": ln, p = itf8(d, p); p += ln
        else: break
    return out


def parse_comp_header(d):
    """-> {series: (codec, params bytes)}"""
    p = 0
    sz, p = itf8(d, p); p += sz                                   # preservation map
    sz, p = itf8(d, p); end = p + sz
    n, p = itf8(d, p)
    enc = {}
    for _ in range(n):
        key = bytes(d[p:p + 2]).decode(); p += 2
        codec, p = itf8(d, p); ln, p = itf8(d, p)
        enc[key] = (codec, bytes(d[p:p + ln])); p += ln
    return enc


class Blk(tuple):
    """(method, content_type, content_id, csize, usize, payload) + .hdr (the block header bytes) + .crc (the stored CRC-32)"""
    def __new__(cls, t, hdr, crc):
        o = super().__new__(cls, t)
        o.hdr, o.crc = hdr, crc
        return o


def containers(b):
    p = 26
    while p < len(b):
        clen = struct.unpack_from("<i", b, p)[0]; p += 4
        refid, p = itf8(b, p); start, p = itf8(b, p); span, p = itf8(b, p); nrec, p = itf8(b, p)
        cnt, p = ltf8(b, p); bases, p = ltf8(b, p)
        nblk, p = itf8(b, p); nland, p = itf8(b, p)
        for _ in range(nland): _, p = itf8(b, p)
        p += 4
        end = p + clen; q = p; blks = []
        while q < end and len(blks) < nblk:
            q0 = q
            method, ctype = b[q], b[q + 1]; q += 2
            cid, q = itf8(b, q); csz, q = itf8(b, q); usz, q = itf8(b, q)
            blks.append(Blk((method, ctype, cid, csz, usz, bytes(b[q:q + csz])), bytes(b[q0:q]), struct.unpack_from("<I", b, q + csz)[0])); q += csz + 4
        yield nrec, blks
        p = end


class Rec(tuple):
    """(name, qual text) + .flag .pos .seqlen .aux ({tag: (type, value)} for one-byte integer tags)"""
    def __new__(cls, name, qual, flag, pos, seqlen, aux, clips=b"", tlen=0):
        o = super().__new__(cls, (name, qual))
        o.flag, o.pos, o.seqlen, o.aux = flag, pos, seqlen, aux
        o.clips, o.tlen = clips, tlen                               # clips: list of soft-clipped base runs, in read order
        return o


def soft_clips(cigar_ops, seq):
    """[(op char, length)] + the read's bases -> the soft-clipped runs (what the SC series stores, one per S op)"""
    out, at = [], 0
    for op, ln in cigar_ops:
        if op == "S": out.append(seq[at:at + ln].encode())
        if op in "MIS=X": at += ln
    return out


def sam_records(path):
    if path.endswith(".bam"):
        return bam_records(path)
    recs = []
    for ln in open(path):
        if ln.startswith("@"): continue
        f = ln.rstrip("\n").split("\t")
        aux = {}
        for t in f[11:]:
            tag, ty, val = t.split(":", 2)
            if ty == "Z": aux[tag] = ("Z", val.encode("latin1"))
        ops = [] if f[5] == "*" else [(m[1], int(m[0])) for m in re.findall(r"(\d+)([MIDNSHP=X])", f[5])]
        recs.append(Rec(f[0], f[10], int(f[1]), int(f[3]), 0 if f[9] == "*" else len(f[9]), aux,
                        soft_clips(ops, f[9]), int(f[8])))
    return recs


def bam_records(path):
    import gzip
    d = gzip.open(path, "rb").read()                              # BGZF is a multi-member gzip file
    assert d[:4] == b"BAM\1"
    p = 8 + struct.unpack_from("<i", d, 4)[0]
    nref = struct.unpack_from("<i", d, p)[0]; p += 4
    for _ in range(nref):
        ln = struct.unpack_from("<i", d, p)[0]; p += 4 + ln + 4
    recs = []
    while p < len(d):
        bs = struct.unpack_from("<i", d, p)[0]; q = p + 4; p = q + bs
        _, pos, lname, _, _, ncig, flag, lseq = struct.unpack_from("<iiBBHHHi", d, q)
        tlen = struct.unpack_from("<i", d, q + 28)[0]
        cig = struct.unpack_from("<%dI" % ncig, d, q + 32 + lname)
        packed = d[q + 32 + lname + 4 * ncig:q + 32 + lname + 4 * ncig + (lseq + 1) // 2]
        seq = "".join("=ACMGRSVTWYHKDBN"[(packed[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(lseq))
        name = d[q + 32:q + 32 + lname - 1].decode()
        a = q + 32 + lname + 4 * ncig + (lseq + 1) // 2
        qual = d[a:a + lseq]
        qtxt = "*" if lseq == 0 or qual[0] == 0xFF else bytes(c + 33 for c in qual).decode("latin1")
        a += lseq
        aux = {}
        size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
        while a < p:
            tag, ty = d[a:a + 2].decode(), chr(d[a + 2]); a += 3
            if ty in ("c", "C"): aux[tag] = (ty, d[a])
            if ty == "Z": aux[tag] = ("Z", d[a:d.index(b"\0", a)])
            if ty in size: a += size[ty]
            elif ty in ("Z", "H"): a = d.index(b"\0", a) + 1
            elif ty == "B":
                sub = chr(d[a]); cnt = struct.unpack_from("<i", d, a + 1)[0]; a += 5 + cnt * size[sub]
            else: raise ValueError(ty)
        recs.append(Rec(name, qtxt, flag, pos + 1, lseq, aux,
                        soft_clips([("MIDNSHP=X"[c & 15], c >> 4) for c in cig], seq), tlen))
    return recs


def main():
    os.makedirs(OUT, exist_ok=True)
    man, nfile = {}, 0
    for cram, sam in PAIRS:
        b = open(os.path.join(REF, cram), "rb").read()
        assert b[:4] == b"CRAM" and b[4] == 3
        recs = sam_records(os.path.join(REF, sam)) if sam else None
        rpos = 0
        for nrec, blks in containers(b):
            if nrec == 0 or not blks: continue
            assert blks[0][1] == 1 and blks[0][0] == 0, "compression header expected raw"
            enc = parse_comp_header(blks[0][5])
            pres = parse_preservation(blks[0][5])
            slice_start = None                                     # alignment start of the (single) slice of this container
            nslices = sum(1 for k in blks if k[1] in (2, 3))
            for k in blks:
                if k[1] in (2, 3) and k[0] == 0 and nslices == 1:
                    _, q = itf8(k[5], 0); slice_start, _ = itf8(k[5], q)
            series_of = {}
            for key, (codec, par) in enc.items():
                if codec == 1: series_of[itf8(par, 0)[0]] = (key, None)             # EXTERNAL
                elif codec == 5: series_of[itf8(par, 1)[0]] = (key, par[0])         # BYTE_ARRAY_STOP
            mine = recs[rpos:rpos + nrec] if recs else None
            rpos += nrec
            for (method, ctype, cid, csz, usz, data) in blks:
                if method != 4 or ctype != 4 or csz == 0: continue
                key, stop = series_of.get(cid, ("??", None))
                exp = None
                if mine is not None:
                    if key == "QS": exp = b"".join(bytes(c - 33 for c in q.encode()) for _, q in mine if q != "*")
                    elif key == "RN" and stop is not None: exp = b"".join(nm.encode() + bytes([stop]) for nm, _ in mine)
                    elif key == "BF": exp = b"".join(put_itf8(r.flag) for r in mine)
                    elif key == "RL": exp = b"".join(put_itf8(r.seqlen) for r in mine)
                    elif key == "AP" and nslices == 1:
                        if pres.get("AP", 1) == 0: exp = b"".join(put_itf8(r.pos) for r in mine)
                        elif slice_start is not None:
                            prev, parts = slice_start, []
                            for r in mine: parts.append(put_itf8(r.pos - prev)); prev = r.pos
                            exp = b"".join(parts)
                    elif key == "??" and cid >= 0x410000:           # aux tag block: content id = tag << 8 | type
                        tag, ty = bytes([(cid >> 16) & 0xFF, (cid >> 8) & 0xFF]).decode("latin1"), chr(cid & 0xFF)
                        if ty in ("c", "C") and all(tag in r.aux for r in mine): exp = bytes(r.aux[tag][1] for r in mine)
                        # a string tag is a BYTE_ARRAY_LEN whose length and bytes share the block: ITF8(len + 1), value, NUL
                        if ty == "Z": exp = b"".join(put_itf8(len(r.aux[tag][1]) + 1) + r.aux[tag][1] + b"\0"
                                                     for r in mine if r.aux.get(tag, ("", 0))[0] == "Z")
                    elif key == "SC" and stop is not None:
                        exp = b"".join(c + bytes([stop]) for r in mine for c in r.clips)
                    elif key == "TS":
                        exp = b"".join(put_itf8(r.tlen & 0xFFFFFFFF) for r in mine)
                    if exp is not None and len(exp) != usz: exp = None
                name = f"{cram.replace('#', '_')}.{nfile}.bin"; nfile += 1
                open(os.path.join(OUT, name), "wb").write(data)
                man[name] = {"source": "test/" + cram, "content_id": cid, "series": key, "order": data[0], "csize": csz,
                             "usize": usz, "expected_hex": exp.hex() if exp is not None and len(exp) <= 4096 else None}
                if exp is not None and len(exp) > 4096:             # a long plaintext travels zlib-packed beside the stream
                    open(os.path.join(OUT, name + ".plain.z"), "wb").write(zlib.compress(exp, 9))
                    man[name]["expected_z"] = name + ".plain.z"
    json.dump(man, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    pinned = sum(1 for v in man.values() if v["expected_hex"] is not None or "expected_z" in v)
    print(len(man), "rANS 4x8 blocks,", pinned, "with SAM-derived plaintext;", "orders", sorted({v['order'] for v in man.values()}))


if __name__ == "__main__":
    main()
