// cram_encode_core.h -- the ENCODE side of the CRAM record layer: BAM records -> the data series of a slice, in the column
// form of the data-parallel decoder (cram_records_fast.h).  Reference: cram_encode_slice / cram_encode_slice_read (cram/cram_encode.c:572-793,
// 1096-1209) writing what process_one_read (:3382-3700) derived from each BAM record -- here both steps are one per-record walk.
//
// Every record is independent once it knows WHERE its values go: a counting walk gives the bytes a record adds to each series, exclusive
// prefix sums per slice give the offsets, the same walk run again writes.  One lane per record for both walks; written once for host
// and device (the CPU harness tests/native/cram_records_host.cpp runs it from plain loops).
//
// Choices the format leaves to the writer, made the simple way (all valid CRAM 3.0, all decoded by the data-parallel passes):
//   * every series EXTERNAL in a block of its own (content id 10 + series), RN / IN / SC as BYTE_ARRAY_STOP (NUL / tab);
//   * every record DETACHED (CF = 2 [+1 when it has bases: the qualities are kept, 0xff ones too]): mate reference, position and template length are stored, no in-slice
//     mate links (htslib links mates inside a slice to save a few bytes per pair; the decoder's output is the same);
//   * AP as a delta from the previous record (first from the slice's start), also in multi-reference slices;
//   * substitution matrix = the default one; a base the code cannot express, or one outside the reference, is a 'B' feature;
//   * tags as stored in BAM, except RG:Z (it becomes the RG series, cram_encode.c:2683-2700): Z / H -> BYTE_ARRAY_STOP(tab) keeping the
//     NUL, fixed-size types -> BYTE_ARRAY_LEN(constant, EXTERNAL), B arrays -> BYTE_ARRAY_LEN(EXTERNAL length, EXTERNAL bytes) in TWO
//     blocks; MD:Z / NM are kept as stored (the decoder then does not regenerate them).
// Not covered (the slice is refused with -3): records with a CIGAR but no bases (CF_NO_SEQ), more than ENC_MAX_TAGS distinct tags or
// ENC_MAX_LINES distinct tag lists in a slice.
#pragma once
#include "cram_records_core.h"

namespace hgr {

// series written (every one an EXTERNAL block of its own); order = content id - 10
enum { W_BF, W_CF, W_RI, W_RL, W_AP, W_RG, W_RN, W_MF, W_NS, W_NP, W_TS, W_TL, W_FN, W_FC, W_FP, W_DL, W_BA, W_BS, W_IN, W_SC, W_HC, W_PD, W_RS, W_MQ, W_QS, W_N };
constexpr int ENC_MAX_TAGS = 64, ENC_MAX_LINES = 256, ENC_KEY_SLOTS = 128, ENC_LINE_SLOTS = 512;
constexpr uint32_t ENC_EMPTY = 0xffffffffu;

struct EncRef { uint64_t off; int64_t len; };                  // reference i: bases at data + off (upper case), len bases; len = 0: not supplied

HGR_FN uint32_t itf8_size(int32_t sv) { const uint32_t v = (uint32_t)sv; return v < 0x80u ? 1u : v < 0x4000u ? 2u : v < 0x200000u ? 3u : v < 0x10000000u ? 4u : 5u; }
HGR_FN uint32_t itf8_write(uint8_t *o, int32_t sv) {          // itf8_put (cram_io.c:277-305)
    const uint32_t v = (uint32_t)sv;
    if (v < 0x80u) { o[0] = (uint8_t)v; return 1; }
    if (v < 0x4000u) { o[0] = (uint8_t)(0x80u | (v >> 8)); o[1] = (uint8_t)v; return 2; }
    if (v < 0x200000u) { o[0] = (uint8_t)(0xc0u | (v >> 16)); o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)v; return 3; }
    if (v < 0x10000000u) { o[0] = (uint8_t)(0xe0u | (v >> 24)); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v; return 4; }
    o[0] = (uint8_t)(0xf0u | (v >> 28)); o[1] = (uint8_t)(v >> 20); o[2] = (uint8_t)(v >> 12); o[3] = (uint8_t)(v >> 4); o[4] = (uint8_t)(v & 0x0fu);
    return 5;
}
HGR_FN uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
HGR_FN uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

// One BAM record (bam_read1's layout behind block_size, sam.c:784-866)
struct BamRec {
    const uint8_t *p; uint32_t bytes;
    int32_t ref_id, pos, mate_ref, mate_pos, tlen; uint32_t l_name, mapq, n_cigar, flag, l_seq;
    const uint8_t *name, *cigar, *seq, *qual, *aux, *end;
};
HGR_FN bool bam_parse(const uint8_t *bam, uint64_t at, uint64_t next, BamRec &R) {
    if (next < at + 36) return false;
    R.p = bam + at + 4; R.bytes = ld32(bam + at);
    if ((uint64_t)R.bytes + 4 != next - at || R.bytes < 32) return false;
    const uint8_t *c = R.p;
    R.ref_id = (int32_t)ld32(c); R.pos = (int32_t)ld32(c + 4); R.l_name = c[8]; R.mapq = c[9]; R.n_cigar = ld16(c + 12); R.flag = ld16(c + 14); R.l_seq = ld32(c + 16);
    R.mate_ref = (int32_t)ld32(c + 20); R.mate_pos = (int32_t)ld32(c + 24); R.tlen = (int32_t)ld32(c + 28);
    R.name = c + 32; R.cigar = R.name + R.l_name; R.seq = R.cigar + 4ull * R.n_cigar; R.qual = R.seq + (R.l_seq + 1u) / 2u; R.aux = R.qual + R.l_seq; R.end = c + R.bytes;
    return R.l_name >= 1 && R.aux <= R.end && (int32_t)R.l_seq >= 0;
}
// bytes of the BAM aux value of this type at p; 0 = malformed
HGR_FN uint32_t aux_size(uint8_t type, const uint8_t *p, const uint8_t *end) {
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'Z': case 'H': { uint32_t n = 0; while (p + n < end && p[n]) n++; return p + n < end ? n + 1u : 0u; }
    case 'B': {
        if (end - p < 5) return 0;
        const uint8_t t = p[0]; const uint32_t es = (t == 'c' || t == 'C') ? 1u : (t == 's' || t == 'S') ? 2u : (t == 'i' || t == 'I' || t == 'f') ? 4u : 0u;
        const uint64_t n = ld32(p + 1);
        return es && 5u + n * es <= (uint64_t)(end - p) ? (uint32_t)(5u + n * es) : 0u;
    }
    default: return 0;
    }
}
HGR_FN bool is_rg(const uint8_t *a) { return a[0] == 'R' && a[1] == 'G' && a[2] == 'Z'; }
HGR_FN uint32_t tag_key(const uint8_t *a) { return (uint32_t)a[0] << 16 | (uint32_t)a[1] << 8 | a[2]; }
HGR_FN uint64_t fnv_step(uint64_t h, uint32_t key) { for (int i = 16; i >= 0; i -= 8) { h ^= (key >> i) & 0xffu; h *= 0x100000001b3ull; } return h; }
constexpr uint64_t FNV0 = 0xcbf29ce484222325ull;

// What the walks need to know about the slice and the batch
struct EncCtx {
    const uint8_t *bam; const uint64_t *rec_off;                    // records of the batch; rec_off[r] .. rec_off[r + 1]
    const uint8_t *data; const EncRef *refs; int32_t nref;          // reference sequences
    const uint8_t *rg_names; const uint32_t *rg_off; int32_t nrg;   // @RG ids back to back
    uint64_t r0; uint32_t nrec;                                     // the slice's records: r0 .. r0 + nrec
    // per-slice tables filled by the tag survey
    const uint32_t *keys; uint32_t nkeys;                           // sorted distinct tag keys (tag << 8 | type)
    const uint64_t *line_hash; uint32_t nlines;                     // hashes of the distinct tag lists, in order of first appearance
    int32_t *fail;                                                  // per slice: -1 malformed BAM record, -3 not covered
};

// an RG:Z tag naming one of the header's read groups becomes the RG series (cram_encode.c:2683-2700); -1: any other tag (an RG:Z value the header
// does not list stays an ordinary tag, so nothing is lost)
HGR_FN int32_t rg_index(const EncCtx &C, const uint8_t *a, uint32_t vs) {
    if (!is_rg(a)) return -1;
    for (int32_t k = 0; k < C.nrg; k++) {
        const uint32_t ln = C.rg_off[k + 1] - C.rg_off[k];
        bool same = ln + 1u == vs;
        for (uint32_t i = 0; same && i < ln; i++) same = C.rg_names[C.rg_off[k] + i] == a[3 + i];
        if (same) return k;
    }
    return -1;
}

// where a walk puts its values: counts (first walk) or bytes (second walk)
template <bool WRITE> struct Sink {
    uint32_t n[W_N];                                                // counting: bytes so far
    uint8_t *p[W_N];                                                // writing: cursors
    // tag k owns columns W_N + 2k (value bytes) and W_N + 2k + 1 (B arrays: the length column).  They stay in memory (the number of tags
    // is not known at compile time): col[c * N + g] holds the record's byte count (counting walk), then its offset into the block (after
    // the prefix sums); out + base[c] is where the slice's block of column c starts
    uint32_t *col; uint64_t N, g; uint8_t *out; const uint64_t *base;
    HGR_FN void itf8(int s, int32_t v) { if (WRITE) p[s] += itf8_write(p[s], v); else n[s] += itf8_size(v); }
    HGR_FN void byte(int s, uint8_t b) { if (WRITE) *p[s]++ = b; else n[s]++; }
    HGR_FN void bytes(int s, const uint8_t *src, uint32_t k) { if (WRITE) { copy_bytes(p[s], src, k); p[s] += k; } else n[s] += k; }
    HGR_FN void tag_bytes(uint32_t c, const uint8_t *src, uint32_t k, int stop) {      // stop >= 0: that byte follows
        uint32_t &cell = col[(uint64_t)c * N + g];
        if (WRITE) { uint8_t *o = out + base[c] + cell; copy_bytes(o, src, k); if (stop >= 0) o[k] = (uint8_t)stop; }
        cell += k + (stop >= 0 ? 1u : 0u);
    }
    HGR_FN void tag_itf8(uint32_t c, int32_t v) {
        uint32_t &cell = col[(uint64_t)c * N + g];
        if (WRITE) cell += itf8_write(out + base[c] + cell, v); else cell += itf8_size(v);
    }
};

HGR_FN int32_t enc_find_key(const uint32_t *keys, uint32_t nkeys, uint32_t key) {      // binary search in the slice's sorted key list
    uint32_t lo = 0, hi = nkeys;
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
    return lo < nkeys && keys[lo] == key ? (int32_t)lo : -1;
}

// The walk of one record.  prev_apos: alignment position of the record before it (slice start for the first).  tag sinks are indexed by the
// key's position in C.keys.  Returns false when the record cannot be encoded (C.fail says why).
// mode: ENC_MODE_MULTI = multi-reference slice (RI stored per record); ENC_MODE_NOREF = the slice is written without any reference (RR = 0).
constexpr int ENC_MODE_MULTI = 1, ENC_MODE_NOREF = 2;
template <bool WRITE>
HGR_FN bool enc_record(const EncCtx &C, uint32_t r, int64_t prev_apos, int mode, Sink<WRITE> &S) {
    const bool multi_ref = (mode & ENC_MODE_MULTI) != 0;
    BamRec B;
    const uint64_t g = C.r0 + r;
    if (!bam_parse(C.bam, C.rec_off[g], C.rec_off[g + 1], B)) { *C.fail = -1; return false; }
    const bool unmapped = (B.flag & BAM_FUNMAP) != 0;
    const int32_t L = (int32_t)B.l_seq;
    // qualities are always kept, also the 0xff bytes of a record that has none -- the reference's choice (cram_encode.c:3763-3790 under CRAM_FLAG_PRESERVE_QUAL_SCORES,
    // which process_one_read sets for every record): a record WITHOUT the flag whose walk meets a 'B' feature comes back with quality 30 everywhere
    // (cram_decode.c:1611-1615, "same as htsjdk"), not with '*'
    const bool has_qual = L > 0;
    if (L == 0 && !unmapped && B.n_cigar) { *C.fail = -3; return false; }                          // CF_NO_SEQ: not covered
    const int64_t apos = (int64_t)B.pos + 1;
    S.itf8(W_BF, (int32_t)B.flag);
    S.itf8(W_CF, CF_DETACHED | (has_qual ? CF_PRESERVE_QUAL : 0));
    if (multi_ref) S.itf8(W_RI, B.ref_id);
    S.itf8(W_RL, L);
    S.itf8(W_AP, (int32_t)(apos - prev_apos));
    // tags: RG:Z -> the RG series; everything else to its tag's block(s); the list of keys picks the dictionary line
    int32_t rg = -1;
    uint64_t lh = FNV0;
    for (const uint8_t *a = B.aux; a < B.end;) {
        if (B.end - a < 3) { *C.fail = -1; return false; }
        const uint32_t vs = aux_size(a[2], a + 3, B.end);
        if (!vs) { *C.fail = -1; return false; }
        const int32_t this_rg = rg_index(C, a, vs);
        if (this_rg >= 0) rg = this_rg;
        else {
            const uint32_t key = tag_key(a);
            lh = fnv_step(lh, key);
            const int32_t k = enc_find_key(C.keys, C.nkeys, key);
            if (k < 0) { *C.fail = -3; return false; }
            const uint32_t vc = (uint32_t)W_N + 2u * (uint32_t)k;
            if (a[2] == 'B') { S.tag_itf8(vc + 1u, (int32_t)vs); S.tag_bytes(vc, a + 3, vs, -1); }      // length to the tag's length block, bytes to its value block
            else if (a[2] == 'Z' || a[2] == 'H') S.tag_bytes(vc, a + 3, vs, '\t');                       // value with its NUL, then the stop byte
            else S.tag_bytes(vc, a + 3, vs, -1);
        }
        a += 3u + vs;
    }
    S.itf8(W_RG, rg);
    S.bytes(W_RN, B.name, B.l_name);                                                             // with the NUL: the stop byte of RN
    S.itf8(W_MF, ((B.flag & BAM_FMREVERSE) ? CRAM_M_REVERSE : 0) | ((B.flag & BAM_FMUNMAP) ? CRAM_M_UNMAP : 0));
    S.itf8(W_NS, B.mate_ref); S.itf8(W_NP, B.mate_pos + 1); S.itf8(W_TS, B.tlen);
    {
        int32_t line = -1;
        if (lh == 0) lh = 1;
        for (uint32_t i = 0; i < C.nlines; i++) if (C.line_hash[i] == lh) { line = (int32_t)i; break; }
        if (line < 0) { *C.fail = -3; return false; }
        S.itf8(W_TL, line);
    }
    auto base_at = [&](int32_t i) -> uint8_t { const uint32_t c = (B.seq[i >> 1] >> ((~i & 1) << 2)) & 15u; return (uint8_t)"=ACMGRSVTWYHKDBN"[c]; };
    if (unmapped) {
        for (int32_t i = 0; i < L; i++) S.byte(W_BA, base_at(i));
        if (has_qual) S.bytes(W_QS, B.qual, (uint32_t)L);
        return true;
    }
    // features from CIGAR + bases + reference (process_one_read's feature generation, cram_encode.c:3480-3680)
    const EncRef *ref = !(mode & ENC_MODE_NOREF) && B.ref_id >= 0 && B.ref_id < C.nref && C.refs[B.ref_id].len > 0 ? &C.refs[B.ref_id] : nullptr;
    int32_t sp = 1, prev = 0, nfeat = 0; int64_t rp = apos;
    auto feature = [&](uint8_t code) { S.byte(W_FC, code); S.itf8(W_FP, sp - prev); prev = sp; nfeat++; };
    for (uint32_t c = 0; c < B.n_cigar; c++) {
        const uint32_t cw = ld32(B.cigar + 4u * c), op = cw & 15u; const int32_t ol = (int32_t)(cw >> 4);
        switch (op) {
        case 0: case 7: case 8:                                                                  // M, =, X: compared with the reference base by base
            if (sp - 1 + ol > L) { *C.fail = -1; return false; }
            for (int32_t i = 0; i < ol; i++, sp++, rp++) {
                const uint8_t b = base_at(sp - 1);
                if (ref && rp >= 1 && rp <= ref->len) {
                    const uint8_t rb = C.data[ref->off + (uint64_t)(rp - 1)];
                    if (b == rb) continue;
                    const int l1 = rb == 'A' ? 0 : rb == 'C' ? 1 : rb == 'G' ? 2 : rb == 'T' ? 3 : 4;
                    const char *sm = l1 == 0 ? "CGTN" : l1 == 1 ? "AGTN" : l1 == 2 ? "ACTN" : l1 == 3 ? "ACGN" : "ACGT";
                    int code = -1;
                    for (int k = 0; k < 4; k++) if ((uint8_t)sm[k] == b) code = k;
                    if (code >= 0) { feature('X'); S.byte(W_BS, (uint8_t)code); continue; }
                }
                feature('B'); S.byte(W_BA, b); S.byte(W_QS, B.qual[sp - 1]);
            }
            break;
        case 1: case 4: {                                                                        // I, S: the bases, then the stop byte
            if (sp - 1 + ol > L) { *C.fail = -1; return false; }
            const int s = op == 1 ? W_IN : W_SC;
            feature(op == 1 ? 'I' : 'S');
            for (int32_t i = 0; i < ol; i++) S.byte(s, base_at(sp - 1 + i));
            S.byte(s, '\t'); sp += ol;
            break;
        }
        case 2: feature('D'); S.itf8(W_DL, ol); rp += ol; break;
        case 3: feature('N'); S.itf8(W_RS, ol); rp += ol; break;
        case 5: feature('H'); S.itf8(W_HC, ol); break;
        case 6: feature('P'); S.itf8(W_PD, ol); break;
        default: *C.fail = -1; return false;
        }
    }
    if (sp - 1 != L) { *C.fail = -1; return false; }
    // FN precedes the features in the decoder's reading order, but it is a series of its own: the order of WRITING does not matter
    S.itf8(W_FN, nfeat);
    S.itf8(W_MQ, (int32_t)B.mapq);
    if (has_qual) S.bytes(W_QS, B.qual, (uint32_t)L);
    return true;
}

// ---- tag survey: which tag keys and which tag lists does the slice hold?  Open-addressing tables per slice (device: atomics). ----
struct EncSurvey {
    uint32_t *keys;            // [slice][ENC_KEY_SLOTS]: tag << 8 | type, ENC_EMPTY = free
    uint64_t *lhash;           // [slice][ENC_LINE_SLOTS]: list hashes, 0 = free
    uint64_t *lcheck;          // [slice][ENC_LINE_SLOTS]: a SECOND, independent 64-bit hash of the list that owns the slot (0 = not yet written): two lists that
                               // collide on lhash would share a dictionary line and decode with each other's tag names -- they would have to collide on
                               // both (128 bits) to pass unnoticed; a mismatch fails the slice with -3
    uint32_t *lfirst;          // [slice][ENC_LINE_SLOTS]: lowest record of the slice with that list
};
HGR_FN void enc_survey_record(const EncCtx &C, uint32_t r, const EncSurvey &V, uint32_t slice) {
    BamRec B;
    const uint64_t g = C.r0 + r;
    if (!bam_parse(C.bam, C.rec_off[g], C.rec_off[g + 1], B)) { *C.fail = -1; return; }
    uint64_t lh = FNV0, lc = 0x9e3779b97f4a7c15ull;
    for (const uint8_t *a = B.aux; a < B.end;) {
        if (B.end - a < 3) { *C.fail = -1; return; }
        const uint32_t vs = aux_size(a[2], a + 3, B.end);
        if (!vs) { *C.fail = -1; return; }
        if (rg_index(C, a, vs) < 0) {
            const uint32_t key = tag_key(a);
            lh = fnv_step(lh, key);
            lc = (lc ^ key) * 0xff51afd7ed558ccdull; lc ^= lc >> 29;                              // (a multiply-xorshift chain: nothing in common with FNV-1a)
            uint32_t *T = V.keys + (size_t)slice * ENC_KEY_SLOTS;
            uint32_t h = (key * 2654435761u) >> 25;                                              // 7 bits
            bool placed = false;
            for (int probe = 0; probe < ENC_KEY_SLOTS; probe++, h = (h + 1u) & (ENC_KEY_SLOTS - 1u)) {
#if defined(__HIP_DEVICE_COMPILE__)
                const uint32_t old = atomicCAS(T + h, ENC_EMPTY, key);
#else
                const uint32_t old = T[h]; if (old == ENC_EMPTY) T[h] = key;
#endif
                if (old == ENC_EMPTY || old == key) { placed = true; break; }
            }
            if (!placed) { *C.fail = -3; return; }
        }
        a += 3u + vs;
    }
    uint64_t *H = V.lhash + (size_t)slice * ENC_LINE_SLOTS; uint32_t *F = V.lfirst + (size_t)slice * ENC_LINE_SLOTS;
    uint32_t h = (uint32_t)(lh >> 40) & (ENC_LINE_SLOTS - 1u);
    if (lh == 0) lh = 1;
    if (lc == 0) lc = 1;
    uint64_t *K = V.lcheck + (size_t)slice * ENC_LINE_SLOTS;
    for (int probe = 0; probe < ENC_LINE_SLOTS; probe++, h = (h + 1u) & (ENC_LINE_SLOTS - 1u)) {
#if defined(__HIP_DEVICE_COMPILE__)
        const unsigned long long old = atomicCAS((unsigned long long *)H + h, 0ull, (unsigned long long)lh);
        if (old == 0ull || old == lh) {
            const unsigned long long oc = atomicCAS((unsigned long long *)K + h, 0ull, (unsigned long long)lc);
            if (oc != 0ull && oc != lc) { *C.fail = -3; return; }                                   // two different tag lists, one hash
            atomicMin(F + h, r);
            return;
        }
#else
        if (H[h] == 0) H[h] = lh;
        if (H[h] == lh) {
            if (K[h] == 0) K[h] = lc;
            if (K[h] != lc) { *C.fail = -3; return; }
            if (r < F[h]) F[h] = r;
            return;
        }
#endif
    }
    *C.fail = -3;
}
}  // namespace hgr
