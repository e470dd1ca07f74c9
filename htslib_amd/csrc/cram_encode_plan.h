// cram_encode_plan.h -- host side of the CRAM record ENCODER (cram_encode_core.h): what the tag survey found is turned into the slice's
// sorted key list and tag dictionary, and -- once the walks have run -- into the compression header block (cram_encode_compression_header,
// reference cram/cram_encode.c:2150-2450, tags :2900-3040) and the slice header block (cram_encode_slice_header, :1060-1095).
// Host-only C++; shared by the device launcher (cram_encode.hip) and the CPU test harness.
#pragma once
#include <algorithm>
#include <string>
#include <utility>
#include <vector>
#include "cram_encode_core.h"
#include "hg_md5.h"

namespace hgr {

struct EncSlice {
    uint64_t r0 = 0; uint32_t nrec = 0;
    std::vector<uint32_t> keys;                    // sorted
    std::vector<uint64_t> lhash; std::vector<uint32_t> lfirst;      // tag lists in order of first appearance
    int32_t min_ref = 0, max_ref = 0; int64_t min_pos = 0, max_end = 0;
    int32_t fail = 0;
    // Does the slice lean on reference sequences?  Decided after the survey (enc_ref_policy): yes when every reference its records sit on was handed
    // over.  Otherwise NO base of the slice is compared with a reference (all stored as B features) and the container says RR = 0, so that a reader
    // does not go looking for a sequence nobody had (htslib: "Unable to fetch reference").  md5: digest of the span of a single-reference slice.
    bool use_ref = true; uint8_t md5[16] = {0};
    int walk_mode() const { return (multi() ? ENC_MODE_MULTI : 0) | (use_ref ? 0 : ENC_MODE_NOREF); }
    bool multi() const { return min_ref != max_ref; }
    int32_t ref_id() const { return multi() ? -2 : min_ref; }
    int64_t start() const { return multi() || min_ref < 0 ? 0 : min_pos; }
    int64_t span() const { return multi() || min_ref < 0 ? 0 : max_end - min_pos + 1; }
    int ncols() const { return W_N + 2 * (int)keys.size(); }
};

// the slice's survey tables (one row of EncSurvey each) -> keys / lines
inline void enc_survey_finish(const uint32_t *keytab, const uint64_t *lh, const uint32_t *lf, EncSlice &S) {
    S.keys.clear(); S.lhash.clear(); S.lfirst.clear();
    for (int i = 0; i < ENC_KEY_SLOTS; i++) if (keytab[i] != ENC_EMPTY) S.keys.push_back(keytab[i]);
    std::sort(S.keys.begin(), S.keys.end());
    std::vector<std::pair<uint32_t, uint64_t>> lines;
    for (int i = 0; i < ENC_LINE_SLOTS; i++) if (lh[i]) lines.emplace_back(lf[i], lh[i]);
    std::sort(lines.begin(), lines.end());
    for (auto &l : lines) { S.lfirst.push_back(l.first); S.lhash.push_back(l.second); }
    if (!S.fail && (S.keys.size() > (size_t)ENC_MAX_TAGS || S.lhash.size() > (size_t)ENC_MAX_LINES)) S.fail = -3;
}

// refs[i].bases / refs[i].len: the sequences the caller has (bases == nullptr: not available).  cram_encode_slice stores the MD5 of the span
// [start, start + span) of a single-reference slice (clamped at the end of the sequence, like the reader's check cram_decode.c:2480-2540).
// (the digest is a pass over the span on the host: enc_ref_md5 can run apart from the decision, beside the device's walks)
template <class Ref>
inline void enc_ref_md5(EncSlice &S, const Ref *refs) {
    memset(S.md5, 0, 16);
    if (S.use_ref && !S.multi() && S.min_ref >= 0 && S.span() > 0) {
        const int64_t len = (int64_t)refs[S.min_ref].len, a = S.start() - 1, b = std::min<int64_t>(a + S.span(), len);
        if (a >= 0 && a < b) md5_of((const uint8_t *)refs[S.min_ref].bases + a, (uint64_t)(b - a), S.md5);
    }
}
template <class Ref>
inline void enc_ref_policy(EncSlice &S, const Ref *refs, int nrefs, bool with_md5 = true) {
    memset(S.md5, 0, 16);
    S.use_ref = true;
    for (int32_t t = std::max<int32_t>(S.min_ref, 0); t <= S.max_ref; t++)
        if (t >= nrefs || !refs[t].bases || !refs[t].len) S.use_ref = false;
    if (with_md5 && S.use_ref && !S.multi() && S.min_ref >= 0 && S.span() > 0) {
        const int64_t len = (int64_t)refs[S.min_ref].len, a = S.start() - 1, b = std::min<int64_t>(a + S.span(), len);
        if (a >= 0 && a < b) md5_of((const uint8_t *)refs[S.min_ref].bases + a, (uint64_t)(b - a), S.md5);
    }
}

namespace enc {
inline void itf8(std::vector<uint8_t> &o, int32_t v) { uint8_t t[5]; const uint32_t n = itf8_write(t, v); o.insert(o.end(), t, t + n); }
inline std::vector<uint8_t> sized(const std::vector<uint8_t> &body) { std::vector<uint8_t> o; itf8(o, (int32_t)body.size()); o.insert(o.end(), body.begin(), body.end()); return o; }
inline void app(std::vector<uint8_t> &o, const std::vector<uint8_t> &x) { o.insert(o.end(), x.begin(), x.end()); }
inline std::vector<uint8_t> external(int32_t cid) { std::vector<uint8_t> p, o; itf8(p, cid); itf8(o, E_EXTERNAL); app(o, sized(p)); return o; }
inline std::vector<uint8_t> stop(uint8_t st, int32_t cid) { std::vector<uint8_t> p{st}, o; itf8(p, cid); itf8(o, E_BYTE_ARRAY_STOP); app(o, sized(p)); return o; }
inline std::vector<uint8_t> constant(int32_t v) { std::vector<uint8_t> p, o; itf8(p, 1); itf8(p, v); itf8(p, 1); itf8(p, 0); itf8(o, E_HUFFMAN); app(o, sized(p)); return o; }      // one symbol, zero bits
inline std::vector<uint8_t> array_len(const std::vector<uint8_t> &len, const std::vector<uint8_t> &val) { std::vector<uint8_t> p, o; app(p, len); app(p, val); itf8(o, E_BYTE_ARRAY_LEN); app(o, sized(p)); return o; }
}  // namespace enc

inline int32_t enc_series_cid(int s) { return 10 + s; }
inline int32_t enc_tag_cid(uint32_t key, bool len_block) { return (int32_t)(key | (len_block ? 1u << 24 : 0u)); }

// Headers of one slice.  tot[c] = bytes of column c (W_N series, then value / length column per key).  blocks: (content id, column) of every
// non-empty block, in the order they are listed in the slice header.
inline void enc_headers(const EncSlice &S, const EncCtx &C, const uint64_t *tot, int64_t record_counter, std::vector<uint8_t> &comp,
                        std::vector<uint8_t> &sh, std::vector<std::pair<int32_t, uint32_t>> &blocks) {
    const uint8_t *bam = C.bam; const uint64_t *rec_off = C.rec_off;     // (host copies: only the first record of every tag list is looked at)
    using namespace enc;
    static const char *NAME[W_N] = {"BF", "CF", "RI", "RL", "AP", "RG", "RN", "MF", "NS", "NP", "TS", "TL", "FN", "FC", "FP", "DL", "BA", "BS", "IN", "SC", "HC", "PD", "RS", "MQ", "QS"};
    comp.clear(); sh.clear(); blocks.clear();
    {   // preservation map: RN, AP, RR, SM (the default matrix), TD
        std::vector<uint8_t> body; itf8(body, 5);
        const uint8_t kv[][3] = {{'R', 'N', 1}, {'A', 'P', 1}, {'R', 'R', (uint8_t)(S.use_ref ? 1 : 0)}};
        for (auto &k : kv) body.insert(body.end(), k, k + 3);
        body.push_back('S'); body.push_back('M'); for (int i = 0; i < 5; i++) body.push_back(0x1B);
        std::vector<uint8_t> td;
        for (size_t l = 0; l < S.lfirst.size(); l++) {                      // the line = the tag keys of its first record, in that record's order (an RG:Z that became the RG series left out)
            BamRec B;
            const uint64_t g = S.r0 + S.lfirst[l];
            if (bam_parse(bam, rec_off[g], rec_off[g + 1], B))
                for (const uint8_t *a = B.aux; a + 3 <= B.end;) { const uint32_t vs = aux_size(a[2], a + 3, B.end); if (!vs) break; if (rg_index(C, a, vs) < 0) td.insert(td.end(), a, a + 3); a += 3u + vs; }
            td.push_back(0);
        }
        if (S.lfirst.empty()) td.push_back(0);
        body.push_back('T'); body.push_back('D'); app(body, sized(td));
        app(comp, sized(body));
    }
    {   // data series encodings
        std::vector<uint8_t> ents; int cnt = 0;
        for (int s = 0; s < W_N; s++) {
            if (s == W_RI && !S.multi()) continue;
            ents.push_back((uint8_t)NAME[s][0]); ents.push_back((uint8_t)NAME[s][1]);
            app(ents, s == W_RN ? stop(0, enc_series_cid(s)) : (s == W_IN || s == W_SC) ? stop('\t', enc_series_cid(s)) : external(enc_series_cid(s)));
            cnt++;
        }
        std::vector<uint8_t> body; itf8(body, cnt); app(body, ents); app(comp, sized(body));
    }
    {   // tag encodings, by type (cram_encode.c:2925-3035)
        std::vector<uint8_t> body; itf8(body, (int32_t)S.keys.size());
        for (uint32_t key : S.keys) {
            const uint8_t t = (uint8_t)key;
            itf8(body, (int32_t)key);
            if (t == 'Z' || t == 'H') app(body, stop('\t', enc_tag_cid(key, false)));
            else if (t == 'B') app(body, array_len(external(enc_tag_cid(key, true)), external(enc_tag_cid(key, false))));
            else app(body, array_len(constant((t == 'A' || t == 'c' || t == 'C') ? 1 : (t == 's' || t == 'S') ? 2 : 4), external(enc_tag_cid(key, false))));
        }
        app(comp, sized(body));
    }
    for (int s = 0; s < W_N; s++) if (tot[s]) blocks.emplace_back(enc_series_cid(s), (uint32_t)s);
    for (size_t k = 0; k < S.keys.size(); k++) {
        if (tot[W_N + 2 * k]) blocks.emplace_back(enc_tag_cid(S.keys[k], false), (uint32_t)(W_N + 2 * k));
        if (tot[W_N + 2 * k + 1]) blocks.emplace_back(enc_tag_cid(S.keys[k], true), (uint32_t)(W_N + 2 * k + 1));
    }
    // slice header (cram_encode_slice_header): reference, start, span, records, record counter, blocks, content ids, embedded reference (none), MD5 of the reference span
    itf8(sh, S.ref_id()); itf8(sh, (int32_t)S.start()); itf8(sh, (int32_t)S.span()); itf8(sh, (int32_t)S.nrec);
    {   // LTF8 record counter
        const uint64_t v = (uint64_t)record_counter;
        if (v < 0x80) sh.push_back((uint8_t)v);
        else if (v < 0x4000) { sh.push_back((uint8_t)(0x80 | (v >> 8))); sh.push_back((uint8_t)v); }
        else if (v < 0x200000) { sh.push_back((uint8_t)(0xc0 | (v >> 16))); sh.push_back((uint8_t)(v >> 8)); sh.push_back((uint8_t)v); }
        else if (v < 0x10000000) { sh.push_back((uint8_t)(0xe0 | (v >> 24))); sh.push_back((uint8_t)(v >> 16)); sh.push_back((uint8_t)(v >> 8)); sh.push_back((uint8_t)v); }
        else { sh.push_back(0xf0 | (uint8_t)((v >> 32) & 0x07)); for (int i = 24; i >= 0; i -= 8) sh.push_back((uint8_t)(v >> i)); }     // 35 bits
    }
    itf8(sh, (int32_t)blocks.size() + 1); itf8(sh, (int32_t)blocks.size());
    for (auto &b : blocks) itf8(sh, b.first);
    itf8(sh, -1); sh.insert(sh.end(), S.md5, S.md5 + 16);
}

}  // namespace hgr
