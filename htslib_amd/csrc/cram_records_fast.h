// cram_records_fast.h -- the DATA-PARALLEL form of the record loop of cram_decode_slice (reference cram/cram_decode.c:2553-2985) for slices in
// which no series shares its bits with another: every series a record reads is EXTERNAL in a block of its own (cram_external_decode_int /
// _char, cram/cram_codecs.c:350-410), a zero-bit HUFFMAN constant (cram_huffman_decode_int0, :2641-2700), a BYTE_ARRAY_STOP over such a block
// (:3586-3624) or a BYTE_ARRAY_LEN with a constant / EXTERNAL length over such blocks (:2937-3010) -- what htslib >= 1.10 writes for
// position-sorted data (cram_encode_compression_header, cram/cram_encode.c:2150-2450; tags :2900-3040).
//
// Why it parallelises: with no shared bit stream, WHICH item of a block a record reads is a count over the records before it --
//   * BF / CF / RL / AP / RG / TL / RI: item r;            MF / NS / NP: the number of detached records before r;  NF, TS, FN / MQ alike;
//   * FC / FP: the sum of FN over the mapped records before r;   BS / DL / IN / SC / BA / QS / HC / PD / RS / BB / QQ: the number of feature
//     codes of that class in all features before the record's first (BA and QS add the whole-read bytes of unmapped / quality-preserving
//     records);
//   * the value of tag k: the number of earlier records whose tag line (TL) lists k.
// So the loop becomes: whole-block column decodes (cram_series.hip), a few rounds of { independent per-record pass -> exclusive prefix sums
// per slice }, and then ONE LANE PER RECORD runs the very same decode_features / decode_body source as the serial chain of
// cram_records_core.h, only with a reader whose cursors start at the record's own prefix sums.  Sizes first (a dry pass counts the CIGAR ops
// and the MD:Z / NM bytes of every record), offsets by prefix sum, then the writing pass; mates by pointer marking (xref, :2140-2307).
//
// Contract with the chain decoder: a slice this path decodes comes out column for column as the chain decodes it.  ANY irregularity --
// a value missing from a block, an out-of-range field, a capacity too small, a mate chain that is not a simple path -- marks the slice
// FAILED here and the launcher hands it to the chain decoder, whose verdict (0, -1 or -3) is the one reported.
//
// Written once for host and device like cram_records_core.h: the kernels of cram_records.hip call these functions with one thread per
// record, tests/native/cram_records_host.cpp calls them from plain loops (test infrastructure; the product exports only the device path).
#pragma once
#include "cram_records_core.h"

namespace hgr {

// how a series is read on this path (resolved per slice by the host: cram_records_fast_plan.h)
enum { F_ABSENT = 0,      // not in the encoding map (or a HUFFMAN code with no symbols): reading it is an error
       F_CONST = 1,       // zero-bit HUFFMAN: k
       F_INT = 2,         // EXTERNAL, ITF8 integers: column `col` of the pool
       F_BYTES = 3,       // EXTERNAL, bytes: data[off .. off + len)
       F_STOP = 4,        // BYTE_ARRAY_STOP: items of data[off .. off + len), item table in column `col` (k = the stop byte)
       F_LENC = 5,        // BYTE_ARRAY_LEN, constant length k (or a scalar EXTERNAL byte codec, k = 1), bytes in data[off .. off + len)
       F_LENV = 6 };      // BYTE_ARRAY_LEN, lengths in column `col`, their running sum in column `col2`, bytes in data[off .. off + len)
struct FSer { int32_t kind, k; uint32_t off, len, col, col2; };
enum { K_BS, K_DL, K_IN, K_SC, K_BA, K_QS, K_HC, K_PD, K_RS, K_BB, K_QQ, NCLS };      // what the features of a record consume
constexpr int FAST_MAX_TAGS = 64;

// the decoded columns: one pool of 32-bit words; column c starts at word col_off[c] and holds col_n[c] values (ITF8 columns), col_n[c] + 1
// item offsets (BYTE_ARRAY_STOP tables) or col_n[c] + 1 running sums (length columns)
struct FView { const uint8_t *data; const uint32_t *pool; const uint64_t *col_off; const uint32_t *col_n; };

HGR_FN bool f_int(const FView &V, const FSer &s, uint32_t idx, int32_t &v) {
    if (s.kind == F_CONST) { v = s.k; return true; }
    if (s.kind != F_INT || idx >= V.col_n[s.col]) return false;
    v = (int32_t)V.pool[V.col_off[s.col] + idx];
    return true;
}
HGR_FN bool f_byte(const FView &V, const FSer &s, uint32_t idx, uint8_t &b) {
    if (s.kind == F_CONST) { b = (uint8_t)s.k; return true; }
    if (s.kind != F_BYTES || idx >= s.len) return false;
    b = V.data[s.off + idx];
    return true;
}
// bytes [a, b) of the block are item idx of an array series
HGR_FN bool f_item(const FView &V, const FSer &s, uint32_t idx, uint32_t &a, uint32_t &b) {
    if (s.kind == F_STOP) {
        if (idx >= V.col_n[s.col]) return false;
        const uint32_t *t = V.pool + V.col_off[s.col];
        a = t[idx]; b = t[idx + 1] - 1u;                                     // the item ends before its stop byte
        return true;
    }
    if (s.kind == F_LENC) {
        const uint64_t e = ((uint64_t)idx + 1u) * (uint32_t)s.k;
        if (s.k < 0 || e > s.len) return false;
        b = (uint32_t)e; a = b - (uint32_t)s.k;
        return true;
    }
    if (s.kind == F_LENV) {
        if (idx >= V.col_n[s.col]) return false;
        const uint32_t *t = V.pool + V.col_off[s.col2];
        a = t[idx]; b = t[idx + 1];
        return b >= a && b <= s.len;                                         // a negative length poisons the sums from there on (0xffffffff)
    }
    return false;
}

// scratch columns of a batch, indexed by the record's number in the batch (rec_off of its slice + r).  Each is filled by a per-record
// pass with a COUNT and turned in place into the exclusive prefix sum over the records of the slice
struct FScr {
    uint32_t *c_det, *c_down, *c_ts, *c_map;  // record is detached / has a mate downstream / reads TS / is mapped  -> index into MF NS NP / NF / TS / FN MQ
    uint32_t *seq_at;                          // read length -> where the record's bases start in the slice's stretch of seq[] / qual[]
    int64_t *ap;                               // AP -> alignment position (inclusive sum from the slice's start when AP is a delta)
    uint32_t *fn;                              // features -> first feature (FC, FP)
    uint32_t *work;                            // features + tags (the chain decoder's run-away guard counts these)
    uint32_t *tag;                             // [tag k][N]: the record has tag k -> index of its value in the tag's block
    uint32_t *cls;                             // [class][N]: items the record consumes -> first item
    uint32_t *aux_stored;                      // bytes of the stored tags (tag + type + value each)
    uint8_t *bits;                             // 1 detached, 2 downstream, 4 reads TS, 8 mapped, 16 MD stored, 32 NM stored
    int32_t *pred;                             // xref: how many records name this one as their mate
    uint64_t N;                                // records in the batch (stride of tag[] and cls[])
};
// (the name lengths / offsets, CIGAR counts / offsets and aux sizes / offsets live in the chain decoder's own scratch columns noff / coff / aoff)

// everything a per-record pass needs about its slice: uniform over the workgroup
struct FCtx {
    const Plan *P;                 // codec_of[] is NOT filled in on this path
    const FSer *ser;               // S_N series, then the plan's distinct tags
    const int32_t *tl_tagidx;      // parallel to P->tl_codec: which of those tags
    uint32_t ntag;
    FView V; FScr Z;
    uint64_t rec_off;
    int32_t nrec, ref_seq_id, nref; int64_t ref_seq_start;
    const RefSpan *refs; int32_t nrefs, decode_md;
    uint32_t cig_cap, name_cap, aux_cap;
    int32_t *fail;                 // the slice's flag: set = the chain decoder takes the slice
    bool want_aux;
};

// one lane's view of the series while it walks ITS record: the questions of cram_records_core.h's Reader, answered from the columns
struct ColReader {
    const Plan *P; const FSer *ser; FView V;
    uint32_t cig_cap_, aux_cap_; int32_t decode_md_;
    int err; uint64_t work;
    uint32_t f_fc, f_fp, imap, c[NCLS];
    HGR_FN bool has(int s) const { return ser[s].kind != F_ABSENT; }
    HGR_FN uint32_t cigar_cap() const { return cig_cap_; }
    HGR_FN uint32_t aux_cap() const { return aux_cap_; }
    HGR_FN int32_t decode_md() const { return decode_md_; }
    HGR_FN const uint8_t *data() const { return V.data; }
    HGR_FN void bulk(uint8_t *dst, const uint8_t *src, uint32_t n, int = 2) { copy_bytes(dst, src, n); }
    HGR_FN void note_access(int, const uint8_t *, uint32_t) {}               // nothing is deferred here: the lane's own accesses are in order
    HGR_FN void bad() { if (!err) err = ERR_MALFORMED; }
    HGR_FN int32_t ival(int s) {
        uint32_t idx;
        switch (s) {
        case S_FN: case S_MQ: idx = imap; break;
        case S_FP: idx = f_fp++; break;
        case S_DL: idx = c[K_DL]++; break;
        case S_HC: idx = c[K_HC]++; break;
        case S_PD: idx = c[K_PD]++; break;
        case S_RS: idx = c[K_RS]++; break;
        default: bad(); return 0;
        }
        int32_t v = 0;
        if (!f_int(V, ser[s], idx, v)) bad();
        return v;
    }
    HGR_FN int32_t bval(int s) {
        uint32_t idx;
        switch (s) {
        case S_FC: idx = f_fc++; break;
        case S_BS: idx = c[K_BS]++; break;
        case S_BA: idx = c[K_BA]++; break;
        case S_QS: idx = c[K_QS]++; break;
        default: bad(); return 0;
        }
        uint8_t b = 0;
        if (!f_byte(V, ser[s], idx, b)) bad();
        return b;
    }
    HGR_FN int32_t array_s(int s, uint8_t *out, uint32_t cap, int) {
        uint32_t idx;
        switch (s) {
        case S_IN: idx = c[K_IN]++; break;
        case S_SC: idx = c[K_SC]++; break;
        case S_BB: idx = c[K_BB]++; break;
        case S_QQ: idx = c[K_QQ]++; break;
        default: bad(); return 0;
        }
        uint32_t a = 0, b = 0;
        if (!f_item(V, ser[s], idx, a, b)) { bad(); return 0; }
        const uint32_t n = b - a;
        if (out) { if (n > cap) { if (!err) err = ERR_UNSUPPORTED; return 0; } copy_bytes(out, V.data + ser[s].off + a, n); }
        return (int32_t)n;
    }
    HGR_FN void bytes_bulk(int s, uint8_t *out, uint32_t n, int) {
        const int k = s == S_QS ? K_QS : K_BA;
        const FSer &e = ser[s];
        if (e.kind == F_CONST) { if (out) for (uint32_t i = 0; i < n; i++) out[i] = (uint8_t)e.k; return; }
        const uint32_t at = c[k];
        if (e.kind != F_BYTES || n > e.len || at > e.len - n) { bad(); return; }
        if (out) copy_bytes(out, V.data + e.off + at, n);
        c[k] = at + n;
    }
};

enum { FB_DET = 1, FB_DOWN = 2, FB_TS = 4, FB_MAPPED = 8, FB_MD = 16, FB_NM = 32 };

// ---- pass 1: flags, read length, AP of record r (nothing depends on other records yet) ----
HGR_FN void fast_m1(const FCtx &C, uint32_t r) {
    const uint64_t g = C.rec_off + r;
    int32_t bf = 0, cf = 0, len = 0, ap = 0;
    bool ok = f_int(C.V, C.ser[S_BF], r, bf);
    ok = f_int(C.V, C.ser[S_CF], r, cf) && ok;
    ok = f_int(C.V, C.ser[S_RL], r, len) && ok;
    ok = f_int(C.V, C.ser[S_AP], r, ap) && ok;
    if (!ok || bf < 0 || bf >= 0x1000 || len < 0) { *C.fail = 1; bf = cf = len = ap = 0; }
    const uint32_t det = (cf & CF_DETACHED) ? 1u : 0u, down = !det && (cf & CF_MATE_DOWNSTREAM) ? 1u : 0u, ts = det || (cf & CF_EXPLICIT_TLEN) ? 1u : 0u,
                   mapped = (bf & BAM_FUNMAP) ? 0u : 1u;
    C.Z.c_det[g] = det; C.Z.c_down[g] = down; C.Z.c_ts[g] = ts; C.Z.c_map[g] = mapped;
    C.Z.bits[g] = (uint8_t)(det * FB_DET | down * FB_DOWN | ts * FB_TS | mapped * FB_MAPPED);
    C.Z.seq_at[g] = (uint32_t)len;
    C.Z.ap[g] = ap;
}

// ---- pass 2 (after the sums of pass 1): features of the record, its name, which tags it has ----
HGR_FN void fast_m2(const FCtx &C, uint32_t r, uint32_t *name_len) {
    const uint64_t g = C.rec_off + r;
    const uint32_t bits = C.Z.bits[g];
    int32_t fn = 0;
    if ((bits & FB_MAPPED) && (!f_int(C.V, C.ser[S_FN], C.Z.c_map[g], fn) || fn < 0)) { *C.fail = 1; fn = 0; }
    C.Z.fn[g] = (uint32_t)fn;
    uint32_t nl = 0;
    if (C.P->rn_included || (bits & FB_DET)) {                               // cram_decode.c:2700-2712, 2745-2757: the name sits with the record, or with the detached mate fields
        uint32_t a = 0, b = 0;
        if (!f_item(C.V, C.ser[S_RN], C.P->rn_included ? r : C.Z.c_det[g], a, b)) *C.fail = 1; else nl = b - a;
    }
    name_len[g] = nl;
    int32_t tl = 0;
    uint32_t ntl = 0;
    for (uint32_t k = 0; k < C.ntag; k++) C.Z.tag[(uint64_t)k * C.Z.N + g] = 0u;
    if (!f_int(C.V, C.ser[S_TL], r, tl) || tl < 0 || tl >= C.P->nTL) *C.fail = 1;
    else {
        for (int32_t t = C.P->tl_off[tl]; t < C.P->tl_off[tl + 1]; t++) {
            const int32_t k = C.tl_tagidx[t];
            if (k < 0) *C.fail = 1; else C.Z.tag[(uint64_t)k * C.Z.N + g] = 1u;
            ntl++;
        }
    }
    C.Z.work[g] = (uint32_t)fn + ntl;
}

// ---- pass 3 (after the sums of pass 2): what the record's features consume of each class; the size of its stored tags ----
HGR_FN void fast_m3(const FCtx &C, uint32_t r) {
    const uint64_t g = C.rec_off + r;
    const uint32_t bits = C.Z.bits[g];
    uint32_t cnt[NCLS];
#pragma unroll
    for (int j = 0; j < NCLS; j++) cnt[j] = 0;
    int32_t fn = 0, len = 0, cf = 0, tl = 0;
    if (bits & FB_MAPPED) (void)f_int(C.V, C.ser[S_FN], C.Z.c_map[g], fn);
    (void)f_int(C.V, C.ser[S_RL], r, len); (void)f_int(C.V, C.ser[S_CF], r, cf);
    if (*C.fail) fn = 0;
    const uint32_t f0 = C.Z.fn[g];
    for (int32_t f = 0; f < fn; f++) {
        uint8_t op = 0;
        if (!f_byte(C.V, C.ser[S_FC], f0 + (uint32_t)f, op)) { *C.fail = 1; break; }
        switch (op) {
        case 'X': cnt[K_BS]++; break;
        case 'D': cnt[K_DL]++; break;
        case 'I': cnt[K_IN]++; break;
        case 'S': cnt[K_SC]++; break;
        case 'i': cnt[K_BA]++; break;
        case 'B': cnt[K_BA]++; cnt[K_QS]++; break;
        case 'Q': cnt[K_QS]++; break;
        case 'H': cnt[K_HC]++; break;
        case 'P': cnt[K_PD]++; break;
        case 'N': cnt[K_RS]++; break;
        case 'b': cnt[K_BB]++; break;
        case 'q': cnt[K_QQ]++; break;
        default: *C.fail = 1; break;
        }
    }
    if (len > 0) {                                                           // the whole-read bytes follow the record's feature bytes (cram_decode.c:2917-2953)
        if (!(bits & FB_MAPPED)) cnt[K_BA] += (uint32_t)len;
        if (cf & CF_PRESERVE_QUAL) cnt[K_QS] += (uint32_t)len;
    }
#pragma unroll
    for (int j = 0; j < NCLS; j++) C.Z.cls[(uint64_t)j * C.Z.N + g] = cnt[j];
    // stored tags (cram_decode_aux): tag, type, value each; cF:C is a note to the decoder and leaves no bytes
    uint32_t stored = 0, extra = 0;
    if (f_int(C.V, C.ser[S_TL], r, tl) && tl >= 0 && tl < C.P->nTL && !*C.fail) {
        for (int32_t t = C.P->tl_off[tl]; t < C.P->tl_off[tl + 1]; t++) {
            const int32_t k = C.tl_tagidx[t], tag = C.P->tl_tag[t];
            if (k < 0) { *C.fail = 1; break; }
            const FSer &e = C.ser[S_N + k];
            uint32_t a = 0, b = 0;
            if (!f_item(C.V, e, C.Z.tag[(uint64_t)k * C.Z.N + g], a, b)) { *C.fail = 1; break; }
            stored += 3u + (b - a);
            if ((tag >> 8) == (('M' << 8) | 'D')) extra |= FB_MD;
            if ((tag >> 8) == (('N' << 8) | 'M')) extra |= FB_NM;
            if (tag == (('c' << 16) | ('F' << 8) | 'C') && b - a == 1u) {
                const uint8_t cF = C.V.data[e.off + a];
                stored -= 4u;
                if (cF & 1) extra |= FB_MD;
                if (cF & 2) extra |= FB_NM;
            }
        }
    }
    C.Z.aux_stored[g] = stored;
    C.Z.bits[g] = (uint8_t)(bits | extra);
}

// the record's reader, positioned by the prefix sums
HGR_FN void fast_reader(const FCtx &C, uint64_t g, ColReader &R) {
    R.P = C.P; R.ser = C.ser; R.V = C.V; R.cig_cap_ = C.cig_cap; R.aux_cap_ = C.aux_cap; R.decode_md_ = C.decode_md;
    R.err = 0; R.work = 0;
    R.f_fc = R.f_fp = C.Z.fn[g]; R.imap = C.Z.c_map[g];
#pragma unroll
    for (int j = 0; j < NCLS; j++) R.c[j] = C.Z.cls[(uint64_t)j * C.Z.N + g];
}
// reference id and position of the record, the reference span it aligns to; false = irregular
HGR_FN bool fast_place(const FCtx &C, uint32_t r, uint64_t g, int32_t &ref_id, int64_t &apos, const RefSpan *&ref) {
    ref_id = C.ref_seq_id;
    if (C.ref_seq_id == -2 && !f_int(C.V, C.ser[S_RI], r, ref_id)) return false;
    if (ref_id < -1 || ref_id >= C.nref) return false;
    apos = C.Z.ap[g];
    if (C.ref_seq_id >= 0 && apos < C.ref_seq_start) return false;
    ref = nullptr;
    for (int32_t i = 0; i < C.nrefs; i++) if (C.refs[i].ref_id == ref_id) { ref = &C.refs[i]; break; }
    if (ref && apos < ref->start) ref = nullptr;
    return true;
}

// ---- pass 4 (after the sums of pass 3): the dry walk -- CIGAR ops and generated aux bytes of the record ----
HGR_FN void fast_m4(const FCtx &C, uint32_t r, uint32_t *ncig_out, uint32_t *aux_out, uint8_t *aux_base) {
    const uint64_t g = C.rec_off + r;
    uint32_t ncig = 0, naux = 0;
    if (!*C.fail) {
        int32_t bf = 0, cf = 0, len = 0, ref_id; int64_t apos; const RefSpan *ref;
        (void)f_int(C.V, C.ser[S_BF], r, bf); (void)f_int(C.V, C.ser[S_CF], r, cf); (void)f_int(C.V, C.ser[S_RL], r, len);
        if (!fast_place(C, r, g, ref_id, apos, ref)) *C.fail = 1;
        else {
            ColReader R; fast_reader(C, g, R);
            Cols O{}; O.aux = C.want_aux ? aux_base : nullptr;
            const uint32_t bits = C.Z.bits[g];
            decode_body<ColReader, true>(R, O, (int)r, bf, cf, len, ref_id, apos, nullptr, nullptr, ref, ncig, naux, (bits & FB_MD) != 0, (bits & FB_NM) != 0, 0u);
            if (R.err) *C.fail = 1;
        }
    }
    ncig_out[g] = ncig;
    aux_out[g] = C.want_aux ? C.Z.aux_stored[g] + naux : 0u;
}

// ---- pass 5 (after the sums of pass 4): the record, written.  O: the slice's columns as the chain decoder sees them (pointers at the
//      slice's first record / its regions of cigar[], names[], aux[]); seq_base: where the slice's bases start in seq[] / qual[] ----
HGR_FN void fast_m5(const FCtx &C, uint32_t r, const Cols &O, uint64_t seq_base) {
    const uint64_t g = C.rec_off + r;
    if (*C.fail) return;
    const uint32_t bits = C.Z.bits[g];
    int32_t bf = 0, cf = 0, len = 0, rg = 0, ref_id; int64_t apos; const RefSpan *ref;
    (void)f_int(C.V, C.ser[S_BF], r, bf); (void)f_int(C.V, C.ser[S_CF], r, cf); (void)f_int(C.V, C.ser[S_RL], r, len);
    if (!fast_place(C, r, g, ref_id, apos, ref) || !f_int(C.V, C.ser[S_RG], r, rg)) { *C.fail = 1; return; }
    O.flags[r] = bf; O.cram_flags[r] = cf; O.ref_id[r] = ref_id; O.len[r] = len; O.apos[r] = apos; O.rg[r] = rg;
    // the name: item r of RN, or the item of the detached record (names come out back to back in record order, like the chain's)
    {
        const uint32_t at = O.name_off[r];
        uint32_t nl = 0;
        if (C.P->rn_included || (bits & FB_DET)) {
            uint32_t a = 0, b = 0;
            if (!f_item(C.V, C.ser[S_RN], C.P->rn_included ? r : C.Z.c_det[g], a, b)) { *C.fail = 1; return; }
            nl = b - a;
            if ((uint64_t)at + nl > C.name_cap) { *C.fail = 1; return; }
            copy_bytes(O.names + at, C.V.data + C.ser[S_RN].off + a, nl);
        }
        O.name_len[r] = (int32_t)nl;
    }
    // mate fields (cram_decode.c:2745-2800)
    int32_t mate_flags = 0, mate_line = -1, mate_ref = -1; int64_t mate_pos = 0, tlen = TLEN_UNSET, etlen = TLEN_UNSET;
    bool ok = true;
    if (bits & FB_DET) {
        int32_t v = 0;
        ok = f_int(C.V, C.ser[S_MF], C.Z.c_det[g], mate_flags) && ok;
        ok = f_int(C.V, C.ser[S_NS], C.Z.c_det[g], mate_ref) && ok;
        if (mate_ref < -1 || mate_ref >= C.nref) ok = false;
        ok = f_int(C.V, C.ser[S_NP], C.Z.c_det[g], v) && ok; mate_pos = v;
        ok = f_int(C.V, C.ser[S_TS], C.Z.c_ts[g], v) && ok; tlen = v;
    } else if (bits & FB_DOWN) {
        int32_t v = 0;
        ok = f_int(C.V, C.ser[S_NF], C.Z.c_down[g], v) && ok; mate_line = (int32_t)((uint32_t)v + r + 1u);
        if (cf & CF_EXPLICIT_TLEN) { ok = f_int(C.V, C.ser[S_TS], C.Z.c_ts[g], v) && ok; etlen = v; }
    } else if (cf & CF_EXPLICIT_TLEN) {
        int32_t v = 0;
        ok = f_int(C.V, C.ser[S_TS], C.Z.c_ts[g], v) && ok; etlen = v;
    }
    if (!ok) { *C.fail = 1; return; }
    O.mate_flags[r] = mate_flags; O.mate_line[r] = mate_line; O.mate_ref_id[r] = mate_ref; O.mate_pos[r] = mate_pos; O.tlen[r] = tlen; O.explicit_tlen[r] = etlen;
    // stored tags
    uint32_t naux = O.aux ? O.aux_off[r] : 0u;
    const uint32_t aux0 = naux;
    if (O.aux) {
        int32_t tl = 0;
        (void)f_int(C.V, C.ser[S_TL], r, tl);
        if ((uint64_t)naux + C.Z.aux_stored[g] > C.aux_cap) { *C.fail = 1; return; }
        for (int32_t t = C.P->tl_off[tl]; t < C.P->tl_off[tl + 1]; t++) {
            const int32_t k = C.tl_tagidx[t], tag = C.P->tl_tag[t];
            const FSer &e = C.ser[S_N + k];
            uint32_t a = 0, b = 0;
            if (!f_item(C.V, e, C.Z.tag[(uint64_t)k * C.Z.N + g], a, b)) { *C.fail = 1; return; }
            if (tag == (('c' << 16) | ('F' << 8) | 'C') && b - a == 1u) continue;
            O.aux[naux] = (uint8_t)(tag >> 16); O.aux[naux + 1] = (uint8_t)(tag >> 8); O.aux[naux + 2] = (uint8_t)tag;
            copy_bytes(O.aux + naux + 3, C.V.data + e.off + a, b - a);
            naux += 3u + (b - a);
        }
        O.aux_len[r] = (int32_t)(naux - aux0);
    }
    // bases, qualities, CIGAR, MD / NM: the chain decoder's own walk
    uint8_t *seq = nullptr, *qual = nullptr;
    if (O.seq) {
        const uint64_t at = seq_base + C.Z.seq_at[g];
        if (at + (uint64_t)len > O.seq_cap) { *C.fail = 1; return; }
        O.seq_off[r] = at; seq = O.seq + at; qual = O.qual + at;
    }
    ColReader R; fast_reader(C, g, R);
    uint32_t ncig = O.cigar_off[r];
    decode_body<ColReader, false>(R, O, (int)r, bf, cf, len, ref_id, apos, seq, qual, ref, ncig, naux, (bits & FB_MD) != 0, (bits & FB_NM) != 0, naux - aux0);
    if (R.err) *C.fail = 1;
}

// ---- mates (cram_decode_slice_xref, cram_decode.c:2140-2307) for slices whose mate links form simple paths: every record is named by at
//      most one other, every link points further down the slice.  Three per-record passes; anything else (unclean) runs the chain decoder's
//      serial xref() on one lane. ----
HGR_FN void fast_xa(const FCtx &C, uint32_t r, const Cols &O, int32_t *unclean) {       // count who names whom (pred[] zeroed before)
    const int32_t ml = O.mate_line[r];
    if (ml == -1) return;
    if (ml <= (int32_t)r || ml >= C.nrec) { *unclean = 1; return; }
#if defined(__HIP_DEVICE_COMPILE__)
    if (atomicAdd(C.Z.pred + C.rec_off + (uint32_t)ml, 1) != 0) *unclean = 1;
#else
    if (C.Z.pred[C.rec_off + (uint32_t)ml]++ != 0) *unclean = 1;
#endif
}
HGR_FN void fast_xb(const FCtx &C, uint32_t r, const Cols &O) {                         // the head of a path walks it: template length of every member
    if (O.mate_line[r] < 0 || C.Z.pred[C.rec_off + r] != 0) return;
    const int32_t rec = (int32_t)r;
    int32_t id2 = rec, ref = O.ref_id[rec], left_cnt = 0, right_cnt = 0;
    int64_t aleft = O.apos[rec], aright = O.aend[rec];
    for (;;) {
        if (aleft > O.apos[id2]) { aleft = O.apos[id2]; left_cnt = 1; } else if (aleft == O.apos[id2]) left_cnt++;
        if (aright < O.aend[id2]) { aright = O.aend[id2]; right_cnt = 1; } else if (aright == O.aend[id2]) right_cnt++;
        if (O.mate_line[id2] == -1) { O.mate_line[id2] = rec; break; }
        id2 = O.mate_line[id2];
        if (O.ref_id[id2] != ref) ref = -1;
    }
    if (ref != -1) {
        int64_t tlen = aright - aleft + 1;
        id2 = rec;
        if (O.apos[id2] == aleft && (O.aend[id2] < aright || left_cnt <= 1)) { O.tlen[id2] = tlen; tlen = -tlen; }
        else if (O.apos[id2] == aleft && O.aend[id2] == aright && left_cnt > 1 && right_cnt > 1) {
            if (O.flags[id2] & BAM_FREAD1) { O.tlen[id2] = tlen; tlen = -tlen; } else O.tlen[id2] = -tlen;
        } else O.tlen[id2] = -tlen;
        id2 = O.mate_line[id2];
        while (id2 != rec) { O.tlen[id2] = tlen; id2 = O.mate_line[id2]; }
    } else {
        id2 = rec;
        O.tlen[id2] = 0;
        id2 = O.mate_line[id2];
        while (id2 != rec) { O.tlen[id2] = 0; id2 = O.mate_line[id2]; }
    }
}
HGR_FN void fast_xc(const FCtx &C, uint32_t r, const Cols &O) {                         // every record: its mate's fields, the mate bits of the flags
    const int32_t rec = (int32_t)r;
    // only bits xref never changes are read from other records (BAM_FUNMAP, BAM_FREVERSE): the passes of the lanes do not depend on each other
    int32_t flags = O.flags[rec];
    if (O.mate_line[rec] >= 0) {
        const int32_t m = O.mate_line[rec];
        O.mate_pos[rec] = O.apos[m];
        O.mate_ref_id[rec] = O.ref_id[m];
        flags |= BAM_FPAIRED;
        const int32_t mf = O.flags[m];
        if (mf & BAM_FUNMAP) { flags |= BAM_FMUNMAP; O.tlen[rec] = 0; }
        if (flags & BAM_FUNMAP) O.tlen[rec] = 0;
        if (mf & BAM_FREVERSE) flags |= BAM_FMREVERSE;
    } else {
        if (O.mate_flags[rec] & CRAM_M_REVERSE) flags |= BAM_FPAIRED | BAM_FMREVERSE;
        if (O.mate_flags[rec] & CRAM_M_UNMAP) flags |= BAM_FMUNMAP;
        if (!(flags & BAM_FPAIRED)) O.mate_ref_id[rec] = -1;
    }
    (void)C;
    O.flags[rec] = flags;
    if (O.tlen[rec] == TLEN_UNSET) O.tlen[rec] = 0;
    if (O.explicit_tlen[rec] != TLEN_UNSET) O.tlen[rec] = O.explicit_tlen[rec];
}

}  // namespace hgr
