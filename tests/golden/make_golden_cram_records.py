#!/usr/bin/env python3
"""Freeze the reference's CRAM v3.0 fixtures as record-decoding vectors (cram_decode_slice).

For every CRAM file of the reference's test directory that has a SAM / BAM twin (test/*.cram, test/tlen/*.cram), each slice
is stored with its DECODED blocks (compression header, slice header, CORE, EXTERNAL blocks by content id -- RAW / gzip / rANS 4x8
payloads are expanded here with zlib and the pinned rANS oracle) and the expected per-record fields taken from the twin WITHOUT
any CRAM code: QNAME, FLAG, reference id, POS, MAPQ, CIGAR, mate reference id, PNEXT, TLEN, SEQ, QUAL, the optional tags as SAM text -- plus the stretch of each
reference the slice's records align to (from the reference's .fa files), which the decoder needs to rebuild the bases.  The tlen/ pairs were written by the
reference's authors to pin the template-length and mate logic of cram_decode_slice_xref (test/tlen/README).

Output: tests/golden/cram_records.json (blocks as base64 of zlib).  Needs /root/reference; run in the build container."""
import base64, glob, gzip, json, os, struct, sys, zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_rans as R                                       # container / block walk, ITF8
from tests import refutil

REF = "/root/reference/test"
FASTA = {"ce#5b_java.cram": "ce.fa", "range.cram": "ce.fa", "auxf#values_java.cram": "auxf.fa", "xx#large_aux_java.cram": "xx.fa"}
TLEN_REF = {"ref": "AAAAACCCCCGGGGGTTTTT"}                          # test/tlen/README
TWINS = {"ce#5b_java.cram": "ce#5b.sam", "auxf#values_java.cram": "auxf#values.sam", "xx#large_aux_java.cram": "xx#large_aux.sam", "range.cram": "range.bam"}


def expand(blk, rans):
    method, _, _, csz, usz, data = blk
    if usz == 0: return b""
    if method == 0: return data
    if method == 1: return zlib.decompress(data, 31)
    if method == 4:
        rc, out = rans.decode(data)
        assert rc == 0 and len(out) == usz
        return out
    raise ValueError("block method %d" % method)


def aux_to_text(b):
    """BAM aux bytes -> SAM text tags (integers of every width print as type i, as samtools view does)"""
    out, p = [], 0
    size = {"c": ("<b", 1), "C": ("<B", 1), "s": ("<h", 2), "S": ("<H", 2), "i": ("<i", 4), "I": ("<I", 4), "f": ("<f", 4)}
    while p < len(b):
        tag, ty = b[p:p + 2].decode("latin1"), chr(b[p + 2]); p += 3
        if ty == "A": out.append("%s:A:%s" % (tag, chr(b[p]))); p += 1
        elif ty in "cCsSiI": fmt, n = size[ty]; out.append("%s:i:%d" % (tag, struct.unpack_from(fmt, b, p)[0])); p += n
        elif ty == "f": out.append("%s:f:%g" % (tag, struct.unpack_from("<f", b, p)[0])); p += 4
        elif ty in "ZH": e = b.index(b"\0", p); out.append("%s:%s:%s" % (tag, ty, b[p:e].decode("latin1"))); p = e + 1
        elif ty == "B":
            sub = chr(b[p]); cnt = struct.unpack_from("<i", b, p + 1)[0]; p += 5
            fmt, n = size[sub]
            vals = [struct.unpack_from(fmt, b, p + k * n)[0] for k in range(cnt)]; p += cnt * n
            out.append("%s:B:%s%s" % (tag, sub, "".join(",%g" % v if sub == "f" else ",%d" % v for v in vals)))
        else: raise ValueError(ty)
    return out


def short_tag(t):
    """long values (the 900 kB ZZ:Z of xx#large_aux) are compared through a digest"""
    import hashlib
    return t if len(t) <= 512 else "%s<sha1=%s,len=%d>" % (t[:5], hashlib.sha1(t.encode("latin1")).hexdigest(), len(t))


def sam_text(path):
    """-> (reference names, [(qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen)])"""
    if path.endswith(".bam"):
        d = gzip.open(path, "rb").read()
        assert d[:4] == b"BAM\1"
        p = 8 + struct.unpack_from("<i", d, 4)[0]
        nref = struct.unpack_from("<i", d, p)[0]; p += 4
        refs = []
        for _ in range(nref):
            ln = struct.unpack_from("<i", d, p)[0]; refs.append(d[p + 4:p + 4 + ln - 1].decode()); p += 4 + ln + 4
        recs = []
        while p < len(d):
            bs = struct.unpack_from("<i", d, p)[0]; q = p + 4; p = q + bs
            tid, pos, lname, mapq, _, ncig, flag, lseq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", d, q)
            name = d[q + 32:q + 32 + lname - 1].decode()
            cig = struct.unpack_from("<%dI" % ncig, d, q + 32 + lname)
            a = q + 32 + lname + 4 * ncig
            packed = d[a:a + (lseq + 1) // 2]
            seq = "".join("=ACMGRSVTWYHKDBN"[(packed[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(lseq)) or "*"
            ql = d[a + (lseq + 1) // 2:a + (lseq + 1) // 2 + lseq]
            qual = "*" if lseq == 0 or ql[0] == 0xFF else bytes(c + 33 for c in ql).decode("latin1")
            recs.append((name, flag, tid, pos + 1, mapq, [(c >> 4, c & 15) for c in cig], mtid, mpos + 1, tlen, seq, qual, [short_tag(t) for t in aux_to_text(d[a + (lseq + 1) // 2 + lseq:p])],
                         base64.b64encode(d[q:a + (lseq + 1) // 2 + lseq]).decode()))     # the record from refID up to the tags, as the reference wrote it
        return refs, recs
    refs, recs = [], []
    for ln in open(path):
        f = ln.rstrip("\n").split("\t")
        if ln.startswith("@"):
            if f[0] == "@SQ": refs.append([x[3:] for x in f if x.startswith("SN:")][0])
            continue
        import re
        cig = [] if f[5] == "*" else [(int(n), "MIDNSHP=X".index(o)) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", f[5])]
        tid = -1 if f[2] == "*" else refs.index(f[2])
        mtid = -1 if f[6] == "*" else tid if f[6] == "=" else refs.index(f[6])
        recs.append((f[0], int(f[1]), tid, int(f[3]), int(f[4]), cig, mtid, int(f[7]), int(f[8]), f[9], f[10], [short_tag(t) for t in f[11:]]))
    return refs, recs


def fasta(path):
    out, name = {}, None
    for ln in open(path):
        if ln.startswith(">"): name = ln[1:].split()[0]; out[name] = []
        else: out[name].append(ln.strip().upper())
    return {k: "".join(v) for k, v in out.items()}


def ref_spans(recs, refs, bases):
    """the stretch of every reference the mapped records of one slice touch: [[ref id, start (1-based), bases, @SQ length]]"""
    lo, hi = {}, {}
    for r in recs:
        if r[1] & 4 or r[2] < 0: continue
        end = r[3] + sum(n for n, op in r[5] if op in (0, 2, 3, 7, 8)) - 1
        lo[r[2]] = min(lo.get(r[2], r[3]), r[3]); hi[r[2]] = max(hi.get(r[2], end), end)
    out = []
    for tid in sorted(lo):
        full = bases[refs[tid]]
        a, b = max(1, lo[tid]), min(len(full), hi[tid] + 5)
        out.append([tid, a, full[a - 1:b], len(full)])
    return out


def main():
    rans = refutil.Rans4x8Oracle()
    pack = lambda b: base64.b64encode(zlib.compress(bytes(b), 9)).decode()
    out = []
    files = sorted(glob.glob(REF + "/tlen/*.cram")) + [os.path.join(REF, k) for k in sorted(TWINS)]
    for path in files:
        base = os.path.basename(path)
        twin = os.path.join(REF, TWINS[base]) if base in TWINS else path[:-5] + ".sam"
        refs, recs = sam_text(twin)
        bases = fasta(os.path.join(REF, FASTA[base])) if base in FASTA else TLEN_REF
        b = open(path, "rb").read()
        assert b[:4] == b"CRAM" and b[4] == 3
        slices, at = [], 0
        cpos_list, q = [], 26                                          # container offsets and landmarks, for the .crai columns
        while q < len(b):
            q0 = q; clen = struct.unpack_from("<i", b, q)[0]; q += 4
            for _ in range(4): _, q = R.itf8(b, q)
            _, q = R.ltf8(b, q); _, q = R.ltf8(b, q)
            _, q = R.itf8(b, q); nland, q = R.itf8(b, q)
            lm = []
            for _ in range(nland): v, q = R.itf8(b, q); lm.append(v)
            q += 4
            cpos_list.append((q0, lm, clen)); q += clen
        ci = -1
        for nrec, blks in R.containers(b):
            ci += 1
            if not blks or blks[0][1] != 1: continue                 # the file-header container, EOF container
            comp = expand(blks[0], rans)
            k = 1; sidx = 0
            while k < len(blks):
                assert blks[k][1] == 2, "slice header expected"
                sh = expand(blks[k], rans)
                p = 0
                _, p = R.itf8(sh, p); _, p = R.itf8(sh, p); _, p = R.itf8(sh, p)
                n, p = R.itf8(sh, p); _, p = R.ltf8(sh, p); nb, p = R.itf8(sh, p)
                body = blks[k + 1:k + 1 + nb]
                core = [x for x in body if x[1] == 5]
                ext = [x for x in body if x[1] == 4]
                assert len(core) == 1 and len(core) + len(ext) == nb
                slices.append({"comp_hdr": pack(comp), "slice_hdr": pack(sh), "core": pack(expand(core[0], rans)),
                               "blocks": [[x[2], pack(expand(x, rans))] for x in ext], "nrec": n,
                               "cpos": cpos_list[ci][0], "landmark": cpos_list[ci][1][sidx],
                               "slice_bytes": sum(len(x.hdr) + x[3] + 4 for x in blks[k:k + 1 + nb]),
                               "refs": [[t, a, pack(sq.encode()), ln] for t, a, sq, ln in ref_spans(recs[at:at + n], refs, bases)],
                               "expect": [list(r) for r in recs[at:at + n]]})
                at += n
                k += 1 + nb; sidx += 1
        assert at == len(recs), (base, at, len(recs))
        crai = gzip.open(path + ".crai", "rt").read() if os.path.exists(path + ".crai") else None
        rgs = []
        if twin.endswith(".sam"): rgs = [[x[3:] for x in ln.rstrip("\n").split("\t") if x.startswith("ID:")][0] for ln in open(twin) if ln.startswith("@RG")]
        else:
            hd = gzip.open(twin, "rb").read(); hl = struct.unpack_from("<i", hd, 4)[0]
            rgs = [[x[3:] for x in ln.split("\t") if x.startswith("ID:")][0] for ln in hd[8:8 + hl].decode().split("\n") if ln.startswith("@RG")]
        full = [[n, pack(bases[n].encode())] for n in refs] if base in FASTA and FASTA[base] != "ce.fa" else None      # whole small references (ce.fa is 1 MB: spans only)
        if base not in FASTA: full = [[n, pack(TLEN_REF[n].encode())] for n in refs]
        out.append({"cram": pack(b), "ref_names": refs, "full_refs": full, "rg": rgs, "crai": crai, "file": "test/" + os.path.relpath(path, REF), "twin": "test/" + os.path.relpath(twin, REF), "major": 3, "nref": len(refs), "slices": slices})
    json.dump(out, open(os.path.join(HERE, "cram_records.json"), "w"), separators=(",", ":"))
    print(len(out), "files,", sum(len(f["slices"]) for f in out), "slices,", sum(s["nrec"] for f in out for s in f["slices"]), "records,",
          os.path.getsize(os.path.join(HERE, "cram_records.json")), "bytes")


if __name__ == "__main__":
    main()
