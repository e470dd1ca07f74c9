"""Synthetic CRAM 3.0 slices of production size for the record decoder (tests / probes / bench.py): what current htslib writes -- every
variable series in its own EXTERNAL block, constants as zero-bit HUFFMAN codes, names and inserted / clipped bases as
BYTE_ARRAY_STOP -- built from reads whose alignment, bases and qualities are known by construction.  This is a WRITER OF TEST
INPUT, not a restatement of cram_encode_slice; the fixtures of the reference pin the decoder, these slices scale it."""
import numpy as np


def put_itf8(v):
    v &= 0xFFFFFFFF
    if v < 0x80: return bytes([v])
    if v < 0x4000: return bytes([0x80 | (v >> 8), v & 0xFF])
    if v < 0x200000: return bytes([0xC0 | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000: return bytes([0xE0 | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([0xF0 | (v >> 28), (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])

SERIES = ["BF", "CF", "RI", "RL", "AP", "RG", "RN", "MF", "NS", "NP", "TS", "NF", "TL", "FN", "FC", "FP", "DL", "BA", "BS", "IN", "SC", "HC", "PD", "RS", "MQ", "QS"]
BASES = b"ACGT"
SM = ["CGTN", "AGTN", "ACTN", "ACGN", "ACGT"]                          # the default substitution matrix (cram_decode.c:207)


def _ltf8(v):
    assert 0 <= v < 0x80
    return bytes([v])


def _map(entries):
    body = put_itf8(len(entries)) + b"".join(entries)
    return put_itf8(len(body)) + body


TAG_LINES = [b"", b"NMcXZZ", b"NMc"]                               # the tag dictionary of slices made with tags=True
NM_ID, XZ_ID = 60, 61


def compression_header(tags=False, names=True):
    """-> (block bytes, {series: content id})"""
    ids = {s: 10 + k for k, s in enumerate(SERIES)}
    td = b"".join(t + b"\0" for t in TAG_LINES) if tags else b"\0"
    pres = _map([b"RN\x01" if names else b"RN\x00", b"AP\x01", b"RR\x01", b"SM" + bytes([0x1B] * 5), b"TD" + put_itf8(len(td)) + td])
    ext = lambda cid: put_itf8(1) + put_itf8(len(put_itf8(cid))) + put_itf8(cid)
    const = lambda v: put_itf8(3) + (lambda p: put_itf8(len(p)) + p)(put_itf8(1) + put_itf8(v) + put_itf8(1) + put_itf8(0))
    stop = lambda c, cid: put_itf8(5) + (lambda p: put_itf8(len(p)) + p)(bytes([c]) + put_itf8(cid))
    enc = []
    for s in SERIES:
        if s == "RG": e = const(-1)
        elif s == "TL": e = ext(ids[s]) if tags else const(0)
        elif s == "RN": e = stop(0, ids[s])
        elif s in ("IN", "SC"): e = stop(ord("\t"), ids[s])
        else: e = ext(ids[s])
        enc.append(s.encode() + e)
    tagmap = []
    if tags:                                                         # NM:c = BYTE_ARRAY_LEN(constant length 1, EXTERNAL bytes); XZ:Z = BYTE_ARRAY_STOP('\t'), value stored with its NUL
        bal = put_itf8(4) + (lambda p: put_itf8(len(p)) + p)(const(1) + ext(NM_ID))
        tagmap = [put_itf8((ord("N") << 16) | (ord("M") << 8) | ord("c")) + bal, put_itf8((ord("X") << 16) | (ord("Z") << 8) | ord("Z")) + stop(ord("\t"), XZ_ID)]
    return pres + _map(enc) + _map(tagmap), ids


def make_slice(rng, nrec, readlen=100, ref_len=None, unmapped_every=37, detached_every=11, tags=False, names=True, record_counter=0):
    """-> a slice dict in the layout Engine.cram_decode_bam takes (and tests/test_cram_records.load_slices() yields) plus "truth": per record (flag base bits, pos, len, cigar, seq, qual)"""
    ref_len = ref_len or (nrec * 8 + 10 * readlen)
    ref = bytes(BASES[i] for i in rng.integers(0, 4, ref_len))
    comp, ids = compression_header(tags, names)
    col = {s: bytearray() for s in SERIES}
    nm_col, xz_col = bytearray(), bytearray()
    truth, pos_prev, start = [], 1, 1
    positions = np.sort(rng.integers(1, ref_len - 3 * readlen, nrec))
    start = int(positions[0]); pos_prev = start
    for r in range(nrec):
        unmapped = unmapped_every and r % unmapped_every == unmapped_every - 1
        pos = int(positions[r])
        flag = (4 if unmapped else 0) | (16 if rng.random() < 0.5 else 0)
        paired_down = (not unmapped) and r % 2 == 0 and r + 1 < nrec and not (unmapped_every and (r + 1) % unmapped_every == unmapped_every - 1) \
            and not (detached_every and r % detached_every == 0)
        detached = detached_every and r % detached_every == 0
        if paired_down: flag |= 1 | 64
        elif r % 2 == 1 and truth and truth[-1]["down"]: flag |= 1 | 128
        cf = 1 | (2 if detached else 0) | (4 if paired_down else 0)
        col["BF"] += put_itf8(flag); col["CF"] += put_itf8(cf); col["RL"] += put_itf8(readlen)
        col["AP"] += put_itf8(pos - pos_prev); pos_prev = pos
        name = ("r%07d" % r).encode()
        if names or (detached_every and r % detached_every == 0): col["RN"] += name + b"\0"      # RN = 0: only detached records store their name (cram_decode.c:2745-2757)
        if detached:
            col["MF"] += put_itf8(0); col["NS"] += put_itf8(-1); col["NP"] += put_itf8(0); col["TS"] += put_itf8(0)
        elif paired_down: col["NF"] += put_itf8(0)
        aux = b""
        if tags:                                                     # the record's tag line and values (cram_decode_aux reads them right after the mate fields)
            tl = int(rng.integers(0, len(TAG_LINES)))
            col["TL"] += put_itf8(tl)
            if tl:
                v = int(rng.integers(0, 128)); nm_col.append(v); aux += b"NMc" + bytes([v])
            if tl == 1:
                z = bytes(rng.integers(65, 91, int(rng.integers(0, 40)), dtype=np.uint8)) + b"\0"
                xz_col += z + b"\t"; aux += b"XZZ" + z
        qual = rng.integers(2, 41, readlen).astype(np.uint8).tobytes()
        if unmapped:
            seq = bytes(BASES[i] for i in rng.integers(0, 4, readlen))
            col["BA"] += seq; col["QS"] += qual
            truth.append({"flag": flag, "pos": pos, "cigar": [], "seq": seq, "qual": qual, "down": False, "name": name, "aux": aux})
            continue
        # features: a leading soft clip, then substitutions / one insertion / one deletion at increasing read positions
        feats, cigar, seq, rp, sp = [], [], bytearray(), pos, 1                # rp: reference position, sp: read position (1-based)
        def match(n):
            nonlocal rp, sp
            if n <= 0: return
            seq.extend(ref[rp - 1:rp - 1 + n]); rp += n; sp += n
            if cigar and cigar[-1][1] == 0: cigar[-1][0] += n
            else: cigar.append([n, 0])
        if rng.random() < 0.2:
            n = int(rng.integers(1, 8)); clip = bytes(BASES[i] for i in rng.integers(0, 4, n))
            feats.append((ord("S"), sp, clip)); seq.extend(clip); sp += n; cigar.append([n, 4])
        for kind in rng.permutation(["X", "X", "I", "D", "X"])[:int(rng.integers(0, 5))]:
            gap = int(rng.integers(1, 12))
            if sp + gap + 8 > readlen: break
            match(gap)
            if kind == "X":
                rb = "ACGT".index(chr(ref[rp - 1])); code = int(rng.integers(0, 3))        # never N
                feats.append((ord("X"), sp, code)); seq.append(ord(SM[rb][code])); rp += 1; sp += 1
                if cigar and cigar[-1][1] == 0: cigar[-1][0] += 1
                else: cigar.append([1, 0])
            elif kind == "I":
                n = int(rng.integers(1, 5)); ins = bytes(BASES[i] for i in rng.integers(0, 4, n))
                feats.append((ord("I"), sp, ins)); seq.extend(ins); sp += n; cigar.append([n, 1])
            else:
                n = int(rng.integers(1, 6)); feats.append((ord("D"), sp, n)); rp += n; cigar.append([n, 2])
        match(readlen - (sp - 1))
        col["FN"] += put_itf8(len(feats))
        prev = 0
        for code, at, val in feats:
            col["FC"].append(code); col["FP"] += put_itf8(at - prev); prev = at
            if code == ord("S"): col["SC"] += val + b"\t"
            elif code == ord("I"): col["IN"] += val + b"\t"
            elif code == ord("X"): col["BS"].append(val)
            else: col["DL"] += put_itf8(val)
        col["MQ"] += put_itf8(int(rng.integers(0, 61))); col["QS"] += qual
        truth.append({"flag": flag, "pos": pos, "cigar": [c[:] for c in cigar], "seq": bytes(seq), "qual": qual, "down": bool(paired_down), "name": name, "aux": aux})
    sh = put_itf8(0) + put_itf8(start) + put_itf8(ref_len - start) + put_itf8(nrec) + _ltf8(record_counter) + put_itf8(len(SERIES) + 1)
    blocks = [(ids[s], bytes(col[s])) for s in SERIES if len(col[s])]
    if tags: blocks += [(NM_ID, bytes(nm_col)), (XZ_ID, bytes(xz_col))]
    sh += put_itf8(len(blocks)) + b"".join(put_itf8(cid) for cid, _ in blocks) + put_itf8(-1) + bytes(16)
    return {"comp_hdr": comp, "slice_hdr": sh, "core": b"", "blocks": blocks, "nrec": nrec, "refs": [(0, 1, ref, ref_len)], "truth": truth,
            "expect": [[t["name"].decode(), 0, 0, 0, 0, [], 0, 0, 0, t["seq"].decode()] for t in truth]}


def bam_from_slices(engine, slices, copies=1):
    """Slices of make_slice() -> ONE coordinate-sorted BAM stream with a header: the records come out of the pinned record decoder
    (engine.cram_decode_bam), slice k's records are moved to a reference of their own ("chr<k+1>" = the slice's random reference), so that
    any CRAM writer sees sorted single-reference runs.  copies > 1 repeats the whole set under further reference names.
    -> (bam bytes incl. header, [names], [sequences], records)"""
    import struct
    from htslib_amd import _native as nat
    n = len(slices); nrec = sum(s["nrec"] for s in slices)
    keep = []
    arr = nat.cram_slice_array(slices, keep)
    readlen = max(len(t["seq"]) for t in slices[0]["truth"][:64])
    recs, rec_off, st = engine.cram_decode_bam(arr, n, 3, 1, [], nrec * readlen + 4096, nrec * (readlen * 2 + 400))
    assert (st == 0).all(), st
    one = bytearray(recs.tobytes())
    ends, at, starts, slice_of = [], 0, [], []
    for k, s in enumerate(slices):
        for _ in range(s["nrec"]):
            starts.append(at); slice_of.append(k)
            at += 4 + struct.unpack_from("<i", one, at)[0]
        ends.append(at)
    assert at == len(one)
    # every copy: the same records under the copy's reference ids (refID at +4, the mate's at +24 where there is one) -- four byte planes written with numpy
    import numpy as np
    base = np.frombuffer(bytes(one), dtype=np.uint8)
    st = np.asarray(starts, dtype=np.int64); sl = np.asarray(slice_of, dtype=np.int64)
    mate = (base[st + 24].astype(np.int64) | base[st + 25].astype(np.int64) << 8 | base[st + 26].astype(np.int64) << 16 | base[st + 27].astype(np.int64) << 24)
    has_mate = mate < (1 << 31)                                              # (a negative int32 = no mate reference)
    body = np.empty(len(one) * copies, dtype=np.uint8)
    for c in range(copies):
        b = body[c * len(one):(c + 1) * len(one)]
        b[:] = base
        rid = c * n + sl
        for sh in range(4):
            plane = ((rid >> (8 * sh)) & 0xff).astype(np.uint8)
            b[st + 4 + sh] = plane
            b[st[has_mate] + 24 + sh] = plane[has_mate]
    body = body.tobytes()
    names = ["chr%d" % (i + 1) for i in range(n * copies)]
    seqs = [slices[i % n]["refs"][0][2] for i in range(n * copies)]
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (a.encode(), len(q)) for a, q in zip(names, seqs))
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(names))
    for a, q in zip(names, seqs): hdr += struct.pack("<i", len(a) + 1) + a.encode() + b"\0" + struct.pack("<i", len(q))
    return bytes(hdr) + bytes(body), names, seqs, nrec * copies
