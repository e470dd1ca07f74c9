// cram_fastpath_proto.cpp -- TEST INFRASTRUCTURE, a CPU prototype of the data-parallel ("EXTERNAL-only") CRAM record decoder that
// DESIGN.md 4.11 / 9 plans for the device: every step below is either a whole-block column decode, a prefix sum, or a map over
// records / features whose iterations are independent -- the shape of a kernel sequence.  It exists to pin the INDEX ARITHMETIC (which
// item of which series a record or feature reads) before any kernel is written: tests/test_cram_records.py checks it, column by
// column, against the pinned chain decoder (cram_records_core.h) on synthetic production-size slices.
//
// Eligible slices: every series a record reads is EXTERNAL in a block no other series reads, a zero-bit HUFFMAN constant, or
// BYTE_ARRAY_STOP over such a block; read names present in every record (RN = 1); tag values BYTE_ARRAY_LEN (constant or EXTERNAL
// length, EXTERNAL bytes) or BYTE_ARRAY_STOP over blocks of their own.  Anything else returns -3 and stays with the chain decoder.
// MD / NM regeneration is not part of the prototype (compare with decode_md = 0).
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../htslib_amd/csrc/cram_records_plan.h"

using namespace hgr;

namespace {

struct Col { bool konst = false; int32_t k = 0; std::vector<int32_t> v; const uint8_t *bytes = nullptr; uint32_t nbytes = 0; std::vector<uint32_t> item; };   // item: BYTE_ARRAY_STOP offsets (n + 1)

template <class T> std::vector<T> exscan(const std::vector<T> &a) { std::vector<T> o(a.size() + 1, 0); for (size_t i = 0; i < a.size(); i++) o[i + 1] = o[i] + a[i]; return o; }   // SCAN

struct out_cols {            // = record_cols of cram_records_host.cpp
    int32_t *flags, *cram_flags, *ref_id, *len, *rg, *mqual, *mate_ref_id, *ncigar, *name_len;
    int64_t *apos, *aend, *mate_pos, *tlen;
    uint64_t *cigar_off, *name_off;
    uint32_t *cigar; uint8_t *names;
    uint64_t *seq_off; uint8_t *seq, *qual;
    uint64_t *aux_off; int32_t *aux_len; uint8_t *aux;
};

}  // namespace

extern "C" int hgr_proto_decode_slice(const SliceIn *in, int major, int nref, size_t cigar_cap, size_t name_cap, size_t seq_cap, size_t aux_cap, const out_cols *out,
                                      uint64_t *used /* cigar, names, seq, aux */) {
    PlanHost H; SliceHeader sh;
    if (plan_from_compression_header(H, in->comp_hdr, in->comp_hdr_len) || parse_slice_header(in->slice_hdr, in->slice_hdr_len, major, sh)) return -1;
    const Plan &P = H.plan;
    if (!P.rn_included || sh.ref_seq_id == -2) return -3;
    // ---- eligibility + STEP 1: whole-block column decodes (hg_cram_itf8_decode_dev / hg_cram_byte_array_stop_dev on the device) ----
    auto block = [&](int32_t slot, const uint8_t *&p, uint32_t &n) { p = nullptr; n = 0; for (uint32_t k = 0; k < in->nblocks; k++) if (in->content_id[k] == H.slot_id[(size_t)slot]) { p = in->data[k]; n = in->len[k]; } };
    std::vector<int> slot_users(H.slot_id.size(), 0);
    static const int ints[] = {S_BF, S_CF, S_RL, S_AP, S_RG, S_MF, S_NS, S_NP, S_TS, S_NF, S_TL, S_FN, S_FP, S_DL, S_HC, S_PD, S_RS, S_MQ};     // TL: which tag line a record has
    static const int bytes_[] = {S_FC, S_BS, S_BA, S_QS};
    static const int arrays[] = {S_RN, S_IN, S_SC};
    Col col[S_N];
    for (int s : ints) {
        const int32_t ci = P.codec_of[s]; if (ci < 0) continue;
        const Codec &C = H.codecs[(size_t)ci];
        if (C.kind == E_HUFFMAN && C.b == 1 && H.huff[(size_t)C.a].len == 0) { col[s].konst = true; col[s].k = H.huff[(size_t)C.a].symbol; continue; }
        if (C.kind != E_EXTERNAL) return -3;
        slot_users[(size_t)C.a]++;
        const uint8_t *p; uint32_t n; block(C.a, p, n);
        Cursor c{p, p + n};
        while (c.p < c.end) { col[s].v.push_back(c.itf8()); if (c.bad) return -1; }                 // MAP over the block (start-finding as in cram_series.hip)
    }
    for (int s : bytes_) {
        const int32_t ci = P.codec_of[s]; if (ci < 0) continue;
        const Codec &C = H.codecs[(size_t)ci];
        if (C.kind != E_EXTERNAL) return -3;
        slot_users[(size_t)C.a]++;
        block(C.a, col[s].bytes, col[s].nbytes);
    }
    for (int s : arrays) {
        const int32_t ci = P.codec_of[s]; if (ci < 0) continue;
        const Codec &C = H.codecs[(size_t)ci];
        if (C.kind != E_BYTE_ARRAY_STOP) return -3;
        slot_users[(size_t)C.a]++;
        block(C.a, col[s].bytes, col[s].nbytes);
        col[s].item.push_back(0);
        for (uint32_t i = 0; i < col[s].nbytes; i++) if (col[s].bytes[i] == (uint8_t)C.b) col[s].item.push_back(i + 1);   // MAP + compaction
    }
    for (int u : slot_users) if (u > 1) return -3;                       // a shared block interleaves its series record by record: chain decoder
    used[3] = 0;
    auto I = [&](int s, size_t i) -> int32_t { return col[s].konst ? col[s].k : col[s].v.at(i); };
    const size_t n = (size_t)sh.nrec;
    // ---- STEP 2: per record, from BF / CF alone ----
    std::vector<uint32_t> det(n), down(n), ts(n), mapped(n), unm(n);
    for (size_t r = 0; r < n; r++) {                                     // MAP
        const int32_t bf = I(S_BF, r), cf = I(S_CF, r);
        det[r] = (cf & CF_DETACHED) != 0; down[r] = !det[r] && (cf & CF_MATE_DOWNSTREAM); ts[r] = det[r] || (cf & CF_EXPLICIT_TLEN) != 0;
        mapped[r] = !(bf & BAM_FUNMAP); unm[r] = !mapped[r];
    }
    const auto i_det = exscan(det), i_down = exscan(down), i_ts = exscan(ts), i_map = exscan(mapped);
    std::vector<int64_t> apos(n);
    { int64_t run = sh.ref_seq_start; for (size_t r = 0; r < n; r++) { run = P.ap_delta ? run + I(S_AP, r) : I(S_AP, r); apos[r] = run; } }   // SCAN (inclusive)
    // ---- STEP 3: features of a record ----
    std::vector<uint32_t> fn(n);
    for (size_t r = 0; r < n; r++) fn[r] = mapped[r] ? (uint32_t)I(S_FN, i_map[r]) : 0u;        // MAP
    const auto f_off = exscan(fn);
    const size_t nf = f_off[n];
    // ---- STEP 4: per feature, which item of which series ----
    std::vector<uint32_t> isX(nf), isD(nf), isI(nf), isS(nf), isBA(nf), isQS(nf), isH(nf), isP(nf), isN(nf);
    for (size_t f = 0; f < nf; f++) {                                    // MAP
        const uint8_t c = col[S_FC].bytes[f];
        isX[f] = c == 'X'; isD[f] = c == 'D'; isI[f] = c == 'I'; isS[f] = c == 'S'; isBA[f] = c == 'i' || c == 'B'; isQS[f] = c == 'Q' || c == 'B'; isH[f] = c == 'H'; isP[f] = c == 'P'; isN[f] = c == 'N';
        if (c == 'b' || c == 'q') return -3;                             // BB / QQ: same scheme, not needed for the inputs at hand
    }
    const auto xX = exscan(isX), xD = exscan(isD), xI = exscan(isI), xS = exscan(isS), xBA = exscan(isBA), xQS = exscan(isQS), xH = exscan(isH), xP = exscan(isP), xN = exscan(isN);
    // bytes of QS / BA before each record: feature bytes of earlier records + their bulk bytes (cram_decode_slice order: features, then bulk)
    std::vector<uint64_t> qs_rec(n), ba_rec(n);
    for (size_t r = 0; r < n; r++) {                                     // MAP (the per-record feature counts are differences of the scans)
        const uint64_t len = (uint64_t)I(S_RL, r); const bool pres = (I(S_CF, r) & CF_PRESERVE_QUAL) != 0;
        qs_rec[r] = (xQS[f_off[r + 1]] - xQS[f_off[r]]) + (pres ? len : 0);
        ba_rec[r] = (xBA[f_off[r + 1]] - xBA[f_off[r]]) + (unm[r] ? len : 0);
    }
    const auto qs_at = exscan(qs_rec), ba_at = exscan(ba_rec);
    std::vector<uint64_t> lens(n); for (size_t r = 0; r < n; r++) lens[r] = (uint64_t)I(S_RL, r);
    const auto seq_at = exscan(lens);                                    // bases / qualities of record r live at seq_at[r] (dense, in record order)
    if (seq_at[n] > seq_cap) return -5;
    // ---- STEP 5: one "lane" per record ----
    const RefIn *ref = nullptr;
    for (uint32_t k = 0; k < in->nrefs; k++) if (in->refs[k].ref_id == sh.ref_seq_id) ref = &in->refs[k];
    std::vector<uint32_t> ncig(n, 0);
    std::vector<std::vector<uint32_t>> cig(n);                           // on the device: count pass, scan, write pass
    std::vector<int64_t> aend(n);
    for (size_t r = 0; r < n; r++) {                                     // MAP
        const int32_t len = I(S_RL, r), cf = I(S_CF, r);
        uint8_t *seq = out->seq + seq_at[r], *qual = out->qual + seq_at[r];
        const RefIn *rf = ref && apos[r] >= ref->start ? ref : nullptr;
        if (!rf) memset(seq, '=', (size_t)len);
        out->seq_off[r] = seq_at[r];
        if (unm[r]) {
            memcpy(seq, col[S_BA].bytes + ba_at[r], (size_t)len);
            if (cf & CF_PRESERVE_QUAL) memcpy(qual, col[S_QS].bytes + qs_at[r], (size_t)len); else memset(qual, 255, (size_t)len);
            aend[r] = apos[r]; out->mqual[r] = 0;
            continue;
        }
        if (!(cf & CF_PRESERVE_QUAL)) memset(qual, 255, (size_t)len);
        int64_t ref_pos = apos[r] - 1; int32_t prev = 0, seq_pos = 1, cl = 0, cop = C_MATCH;
        auto &cg = cig[r];
        auto flush_unless = [&](int op) { if (cl && cop != op) { cg.push_back(((uint32_t)cl << 4) | (uint32_t)cop); cl = 0; } };
        auto copy_ref = [&](int32_t at, int64_t cnt) { if (rf && cnt > 0) memcpy(seq + at, rf->bases + (ref_pos + 1 - rf->start), (size_t)cnt); };
        uint64_t qs_f = qs_at[r], ba_f = ba_at[r];                      // this record's feature bytes come first in QS / BA
        for (size_t f = f_off[r]; f < f_off[r + 1]; f++) {
            const uint8_t op = col[S_FC].bytes[f];
            const int32_t pos = col[S_FP].konst ? col[S_FP].k + prev : col[S_FP].v[f] + prev;
            if (pos > seq_pos) { copy_ref(seq_pos - 1, pos - seq_pos); flush_unless(C_MATCH); cop = C_MATCH; cl += pos - seq_pos; ref_pos += pos - seq_pos; seq_pos = pos; }
            prev = pos;
            switch (op) {
            case 'S': { if (cl) { cg.push_back(((uint32_t)cl << 4) | (uint32_t)cop); cl = 0; }
                        const uint32_t a = col[S_SC].item[xS[f]], b = col[S_SC].item[xS[f] + 1] - 1; memcpy(seq + pos - 1, col[S_SC].bytes + a, b - a);
                        cg.push_back(((b - a) << 4) | C_SOFT_CLIP); cop = C_SOFT_CLIP; seq_pos += (int32_t)(b - a); break; }
            case 'X': { flush_unless(C_MATCH); const int base = col[S_BS].bytes[xX[f]] & 3;
                        if (rf) { const uint8_t rc = rf->bases[ref_pos + 1 - rf->start]; const int l1 = rc == 'A' ? 0 : rc == 'C' ? 1 : rc == 'G' ? 2 : rc == 'T' ? 3 : 4; seq[pos - 1] = H.sm[(size_t)(4 * l1 + base)]; }
                        else seq[pos - 1] = H.sm[(size_t)(16 + base)];
                        cop = C_MATCH; cl++; seq_pos++; ref_pos++; break; }
            case 'D': { flush_unless(C_DEL); const int32_t v = I(S_DL, xD[f]); cop = C_DEL; cl += v; ref_pos += v; break; }
            case 'I': { flush_unless(C_INS); const uint32_t a = col[S_IN].item[xI[f]], b = col[S_IN].item[xI[f] + 1] - 1; memcpy(seq + pos - 1, col[S_IN].bytes + a, b - a);
                        cop = C_INS; cl += (int32_t)(b - a); seq_pos += (int32_t)(b - a); break; }
            case 'i': flush_unless(C_INS); seq[pos - 1] = col[S_BA].bytes[ba_f++]; cop = C_INS; cl++; seq_pos++; break;
            case 'B': flush_unless(C_MATCH); seq[pos - 1] = col[S_BA].bytes[ba_f++]; qual[pos - 1] = col[S_QS].bytes[qs_f++]; cop = C_MATCH; cl++; seq_pos++; ref_pos++; break;
            case 'Q': qual[pos - 1] = col[S_QS].bytes[qs_f++]; break;
            case 'H': { flush_unless(C_HARD_CLIP); cop = C_HARD_CLIP; cl += I(S_HC, xH[f]); break; }
            case 'P': { flush_unless(C_PAD); cop = C_PAD; cl += I(S_PD, xP[f]); break; }
            case 'N': { flush_unless(C_REF_SKIP); const int32_t v = I(S_RS, xN[f]); cop = C_REF_SKIP; cl += v; ref_pos += v; break; }
            default: return -1;
            }
        }
        if (len >= seq_pos) { copy_ref(seq_pos - 1, len - (seq_pos - 1)); ref_pos += len - seq_pos + 1; flush_unless(C_MATCH); cop = C_MATCH; cl += len - seq_pos + 1; }
        if (cl) cg.push_back(((uint32_t)cl << 4) | (uint32_t)cop);
        aend[r] = ref_pos > apos[r] ? ref_pos : apos[r];
        out->mqual[r] = I(S_MQ, i_map[r]);
        if (cf & CF_PRESERVE_QUAL) memcpy(qual, col[S_QS].bytes + qs_f, (size_t)len);      // the bulk qualities follow the record's feature bytes
    }
    for (size_t r = 0; r < n; r++) ncig[r] = (uint32_t)cig[r].size();
    const auto c_off = exscan(ncig);                                     // SCAN, then the write pass
    if (c_off[n] > cigar_cap) return -5;
    std::vector<int32_t> mate_flags(n, 0), mate_line(n, -1); std::vector<int64_t> etlen(n, TLEN_UNSET); std::vector<uint32_t> coff(n), noff(n);
    uint64_t names_used = 0;
    for (size_t r = 0; r < n; r++) {                                     // MAP: the fixed fields, all indices known
        const int32_t cf = I(S_CF, r);
        out->flags[r] = I(S_BF, r); out->cram_flags[r] = cf; out->ref_id[r] = sh.ref_seq_id; out->len[r] = (cf & CF_NO_SEQ) ? 0 : I(S_RL, r); out->rg[r] = I(S_RG, r);
        out->apos[r] = apos[r]; out->aend[r] = aend[r];
        out->mate_pos[r] = 0; out->mate_ref_id[r] = -1; out->tlen[r] = TLEN_UNSET;
        if (det[r]) { mate_flags[r] = I(S_MF, i_det[r]); out->mate_ref_id[r] = I(S_NS, i_det[r]); out->mate_pos[r] = I(S_NP, i_det[r]); out->tlen[r] = I(S_TS, i_ts[r]); }
        else if (down[r]) { mate_line[r] = I(S_NF, i_down[r]) + (int32_t)r + 1; if (cf & CF_EXPLICIT_TLEN) etlen[r] = I(S_TS, i_ts[r]); }
        else if (cf & CF_EXPLICIT_TLEN) etlen[r] = I(S_TS, i_ts[r]);
        memcpy(out->cigar + c_off[r], cig[r].data(), cig[r].size() * 4);
        out->cigar_off[r] = c_off[r]; out->ncigar[r] = (int32_t)ncig[r]; coff[r] = c_off[r];
        const uint32_t a = col[S_RN].item[r], b = col[S_RN].item[r + 1] - 1;      // read name r = item r of the RN block
        if (a + (uint64_t)(b - a) > name_cap) return -5;
        memcpy(out->names + a, col[S_RN].bytes + a, b - a);              // names keep their block offsets: no scan needed
        out->name_off[r] = a; out->name_len[r] = (int32_t)(b - a); noff[r] = a; names_used = b;
    }
    // ---- STEP 5b: tags.  Per tag (a codec of the tag encoding map): which records carry it (their TL line lists it) -> scan -> item index;
    //      where item k of the tag's block starts: k * L (constant length), a scan over its length column, or the stop-byte split ----
    {
        struct Tag { int32_t ci; std::vector<uint64_t> at; const uint8_t *bytes; };      // at: n_items + 1 offsets into bytes; STOP items end one byte before the next start
        std::vector<Tag> tags; std::vector<int> stopped;
        std::vector<int32_t> tag_of((size_t)H.tl_codec.size(), -1);
        for (size_t t = 0; t < H.tl_codec.size(); t++) {
            const int32_t ci = H.tl_codec[t];
            if (ci < 0) return -1;
            size_t k = 0; while (k < tags.size() && tags[k].ci != ci) k++;
            if (k == tags.size()) {
                const Codec &C = H.codecs[(size_t)ci];
                Tag T; T.ci = ci; T.bytes = nullptr; uint32_t nbytes = 0;
                if (C.kind == E_BYTE_ARRAY_STOP) {
                    if (++slot_users[(size_t)C.a] > 1) return -3;
                    block(C.a, T.bytes, nbytes);
                    T.at.push_back(0);
                    for (uint32_t i = 0; i < nbytes; i++) if (T.bytes[i] == (uint8_t)C.b) T.at.push_back((uint64_t)i + 1);
                    stopped.push_back(1);
                } else if (C.kind == E_BYTE_ARRAY_LEN) {
                    const Codec &L = H.codecs[(size_t)C.a], &V = H.codecs[(size_t)C.b];
                    if (V.kind != E_EXTERNAL || ++slot_users[(size_t)V.a] > 1) return -3;
                    block(V.a, T.bytes, nbytes);
                    T.at.push_back(0);
                    if (L.kind == E_HUFFMAN && L.b == 1 && H.huff[(size_t)L.a].len == 0) {
                        const uint64_t len = (uint64_t)H.huff[(size_t)L.a].symbol;
                        if (len == 0) return -3;
                        for (uint64_t a = len; a <= nbytes; a += len) T.at.push_back(a);
                    } else if (L.kind == E_EXTERNAL) {
                        if (++slot_users[(size_t)L.a] > 1) return -3;
                        const uint8_t *lp; uint32_t ln; block(L.a, lp, ln);
                        Cursor c{lp, lp + ln}; uint64_t a = 0;
                        while (c.p < c.end) { a += (uint64_t)c.itf8(); if (c.bad) return -1; T.at.push_back(a); }      // SCAN over the length column
                    } else return -3;
                    stopped.push_back(0);
                } else return -3;
                tags.push_back(T);
            }
            tag_of[t] = (int32_t)k;
        }
        std::vector<std::vector<uint64_t>> item(tags.size());               // item[k][r] = index of record r's value in tag k's block
        std::vector<uint64_t> asz(n, 0);
        std::vector<int32_t> line(n);
        for (size_t r = 0; r < n; r++) { line[r] = I(S_TL, r); if (line[r] < 0 || line[r] >= P.nTL) return -1; }       // MAP
        for (size_t k = 0; k < tags.size(); k++) {
            std::vector<uint64_t> has(n, 0);
            for (size_t r = 0; r < n; r++) for (int32_t t = H.tl_off[(size_t)line[r]]; t < H.tl_off[(size_t)line[r] + 1]; t++) if (tag_of[(size_t)t] == (int32_t)k) has[r] = 1;   // MAP
            item[k] = exscan(has);                                                                                                                                      // SCAN
        }
        auto vlen = [&](size_t k, uint64_t it) -> uint64_t { return tags[k].at.at(it + 1) - tags[k].at.at(it) - (stopped[k] ? 1u : 0u); };
        for (size_t r = 0; r < n; r++) for (int32_t t = H.tl_off[(size_t)line[r]]; t < H.tl_off[(size_t)line[r] + 1]; t++) asz[r] += 3 + vlen((size_t)tag_of[(size_t)t], item[(size_t)tag_of[(size_t)t]][r]);   // MAP
        const auto a_off = exscan(asz);                                                                                                                                  // SCAN
        if (a_off[n] > aux_cap) return -5;
        for (size_t r = 0; r < n && out->aux; r++) {                                                                                                                      // MAP
            uint8_t *o = out->aux + a_off[r];
            for (int32_t t = H.tl_off[(size_t)line[r]]; t < H.tl_off[(size_t)line[r] + 1]; t++) {
                const size_t k = (size_t)tag_of[(size_t)t]; const uint64_t it = item[k][r], ln = vlen(k, it);
                const int32_t tag = H.tl_tag[(size_t)t];
                o[0] = (uint8_t)(tag >> 16); o[1] = (uint8_t)(tag >> 8); o[2] = (uint8_t)tag;
                memcpy(o + 3, tags[k].bytes + tags[k].at[it], (size_t)ln); o += 3 + ln;
            }
            out->aux_off[r] = a_off[r]; out->aux_len[r] = (int32_t)asz[r];
        }
        used[3] = a_off[n];
    }
    // ---- STEP 6: mates (the chain decoder's own function: O(records) of cheap work per slice) ----
    uint32_t totals[4] = {0, 0, 0, 0};
    Cols O{out->flags, out->cram_flags, out->ref_id, out->len, out->rg, out->mqual, mate_flags.data(), out->mate_ref_id, mate_line.data(), out->ncigar, out->name_len,
           coff.data(), noff.data(), out->apos, out->aend, out->mate_pos, out->tlen, etlen.data(), out->cigar, out->names, totals, nullptr, nullptr, nullptr,
           out->seq, out->qual, out->seq_off, nullptr, seq_cap};
    if (xref(O, sh.nrec)) return -1;
    (void)nref;
    used[0] = c_off[n]; used[1] = names_used; used[2] = seq_at[n];
    return 0;
}
